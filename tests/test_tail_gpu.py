"""Target-only tail (shadow_gnn_amd/tail.py): computing the last layers only on the rows the roots depend on
must reproduce the full layer stack -- predictions, loss and every parameter gradient -- because the dropped
rows never reach the loss under residue 'none' + centre pooling (shaDow/layers.py:159-163)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(B, nodes_per, F0, C, seed, deg=3):
    """Block-diagonal batch of B ring-plus-chords subgraphs (symmetric, with self loops, root = first node):
    sparse enough that the 1-hop set of the roots is a small part of the batch."""
    from shadow_gnn_amd import ops
    from shadow_gnn_amd.minibatch import OneBatchSubgraph
    rng = np.random.default_rng(seed)
    blocks = []
    for b in range(B):
        m = nodes_per
        r = np.arange(m)
        rows = np.concatenate([r, r, rng.integers(0, m, deg * m // 2)])
        cols = np.concatenate([r, (r + 1) % m, rng.integers(0, m, deg * m // 2)])
        a = sp.coo_matrix((np.ones(rows.size, np.float32), (rows, cols)), shape=(m, m)).tocsr()
        a = ((a + a.T) > 0).astype(np.float32).tocsr()
        blocks.append(a)
    A = sp.block_diag(blocks, format="csr"); A.sort_indices()
    n = A.shape[0]
    csr = ops.DeviceCSR(torch.from_numpy(A.indptr.astype(np.int32)).to(DEV), torch.from_numpy(A.indices.astype(np.int32)).to(DEV))
    g = torch.Generator(device=DEV).manual_seed(seed)
    feat = torch.randn(n, F0, device=DEV, generator=g)
    tgt = (torch.arange(B) * nodes_per).to(DEV)
    sizes = torch.full((1, B), nodes_per, dtype=torch.int64, device=DEV)
    label = torch.randint(0, C, (B,), device=DEV, generator=g)
    return lambda: OneBatchSubgraph([csr], [feat.clone()], label, sizes, [tgt], [{}]), A, n


def _model(aggr, num_layers, F0, C, dropedge=0.0, dim=64):
    from shadow_gnn_amd.models import DeepGNN
    arch = dict(num_layers=num_layers, num_cls_layers=1, heads=(2 if aggr == "gat" else 1), branch_sharing=False, dim=dim, act="relu",
                layer_norm="norm_feat", feature_augment_ops="sum", aggr=aggr, residue="none", pooling="center",
                loss="softmax", ensemble_act="relu")
    torch.manual_seed(3)
    return DeepGNN(F0, F0, C, 0, arch, [], 1, dict(lr=0.01, dropout=0.0, dropedge=dropedge), "node").to(DEV)


@pytest.mark.parametrize("aggr,num_layers,dropedge", [("sage", 5, 0.0), ("gcn", 3, 0.0), ("sage", 3, 0.1), ("gcn", 5, 0.1),
                                                      ("sage", 1, 0.0), ("gat", 3, 0.0), ("gat", 5, 0.1), ("gat", 1, 0.0)])
def test_pruned_tail_matches_full_stack(aggr, num_layers, dropedge):
    from shadow_gnn_amd.minibatch import TRAIN, VALID
    B, nodes_per, F0, C = 40, 300, 32, 6
    mk, A, n = _batch(B, nodes_per, F0, C, seed=num_layers)
    out = {}
    for prune in (False, True):
        m = _model(aggr, num_layers, F0, C, dropedge)
        m.prune_tail = prune
        e = m.step(VALID, "running", mk())
        torch.manual_seed(17)                      # (same drop-edge mask in both runs)
        m.train()
        bt = mk()
        preds, _ = m(TRAIN, dropedge=dropedge, **bt.to_dict({"feat_ens", "adj_ens", "target_ens", "size_subg_ens", "feat_aug_ens"}))
        loss = m._loss(preds, bt.label)
        loss.backward()
        out[prune] = (e["preds"].detach(), preds.detach(), float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    np.testing.assert_allclose(out[True][0].cpu().numpy(), out[False][0].cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[True][1].cpu().numpy(), out[False][1].cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert abs(out[True][2] - out[False][2]) < 1e-5
    assert out[True][3].keys() == out[False][3].keys()
    for k in out[False][3]:
        a, b = out[True][3][k].cpu().numpy(), out[False][3][k].cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(b).max())), err_msg=k)


def test_tail_plan_sets():
    """The level sets are the BFS balls around the roots in the batch adjacency, renumbered consistently."""
    from shadow_gnn_amd import tail
    B, nodes_per = 30, 400
    mk, A, n = _batch(B, nodes_per, 8, 3, seed=1, deg=1)
    bt = mk()
    csr, tgt = bt.adj_ens[0], bt.target_ens[0]
    levels = tail.build_tail_plan(csr, tgt, 5)
    assert 2 <= len(levels) <= 5
    roots = tgt.cpu().numpy()
    ball = [np.unique(roots)]
    for _ in range(len(levels)):
        prev = ball[-1]
        nb = np.unique(np.concatenate([prev, A[prev].indices]))
        ball.append(nb)
    # top level: rows = roots in target order; below: sorted balls
    top = levels[-1]
    assert np.array_equal(top.rows_full.cpu().numpy(), roots)
    for d, lv in enumerate(reversed(levels)):           # d = 0: last layer
        rows = lv.rows_full.cpu().numpy()
        assert np.array_equal(np.sort(rows), ball[d])
        ins = lv.in_ids_full.cpu().numpy() if lv.in_ids_full is not None else np.arange(n)
        if lv.in_ids_full is not None:
            assert np.array_equal(ins, ball[d + 1])
        # rows of the level = rows of A, columns renumbered into the input numbering
        ip = lv.indptr.cpu().numpy(); ix = lv.indices.cpu().numpy()
        for k in (0, len(rows) // 2, len(rows) - 1):
            assert np.array_equal(np.sort(ins[ix[ip[k]:ip[k + 1]]]), np.sort(A[rows[k]].indices))
            assert ins[lv.self_idx.cpu().numpy()[k]] == rows[k]
    assert levels[0].in_ids_full is None and levels[0].m_in == n
    # transposed level: same edge multiset
    lv = levels[0]
    ti, tx, tp = lv.transposed
    ti, tx = ti.cpu().numpy(), tx.cpu().numpy()
    assert ti[-1] == lv.indices.numel() and np.all(np.diff(ti) >= 0)
    e_fwd = sorted(zip(lv.edge_row.cpu().numpy().tolist(), lv.indices.cpu().numpy().tolist()))
    e_bwd = sorted((int(r), int(c)) for c in range(lv.m_in) for r in tx[ti[c]:ti[c + 1]]) if lv.m_in <= 20000 else None
    if e_bwd is not None:
        assert e_fwd == e_bwd


def test_tail_not_used_when_readout_needs_every_row():
    from shadow_gnn_amd.minibatch import VALID
    from shadow_gnn_amd.models import DeepGNN
    mk, A, n = _batch(10, 100, 16, 4, seed=5)
    for residue, pooling in (("max", "center"), ("none", "mean")):
        arch = dict(num_layers=3, num_cls_layers=1, heads=1, branch_sharing=False, dim=32, act="relu",
                    layer_norm="norm_feat", feature_augment_ops="sum", aggr="sage", residue=residue, pooling=pooling,
                    loss="softmax", ensemble_act="relu")
        torch.manual_seed(0)
        m = DeepGNN(16, 16, 4, 0, arch, [], 1, dict(lr=0.01, dropout=0.0, dropedge=0.0), "node").to(DEV)
        assert not m._tail_prunable(0)
        a = m.step(VALID, "running", mk())["preds"]
        m.prune_tail = True
        b = m.step(VALID, "running", mk())["preds"]
        assert torch.equal(a, b)


@pytest.mark.parametrize("prefetch", [False, True])
def test_minibatch_builds_the_plan_on_its_prefetch_stream(prefetch):
    """MinibatchShallowExtractor.tail_plan_layers: every sampled batch carries its plan; a model stepping with
    it gives the same predictions as the full stack on the same batch."""
    from shadow_gnn_amd.minibatch import TRAIN, VALID, MinibatchShallowExtractor, OneBatchSubgraph
    from shadow_gnn_amd.synthetic import make_graph_numpy
    from shadow_gnn_amd import tail
    N, F0, C = 6000, 20, 7
    indptr, indices = make_graph_numpy(N, 6, seed=4)
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(N, F0, generator=g)
    label = torch.randint(0, C, (N,), generator=g)
    roots = np.random.default_rng(3).permutation(N)[:96]
    mb = MinibatchShallowExtractor.on_device({TRAIN: (indptr, indices)}, {TRAIN: roots},
                                   dict(method="khop", depth=2, budget=6, add_self_edge=True), (), feat, label,
                                   batch_size=32, device=DEV, seed_cpp=11, prefetch=prefetch)
    mb.tail_plan_layers = 4
    mb.epoch_start_reset(0, TRAIN)
    mb.shuffle_entity(TRAIN, perm=np.arange(96))
    m = _model("sage", 4, F0, C)
    for _ in range(3):
        bt = mb.one_batch(TRAIN)
        levels = bt.tail_ens[0]
        ref = tail.build_tail_plan(bt.adj_ens[0], bt.target_ens[0], 4)
        assert len(levels) == len(ref) >= 1
        for a, b in zip(levels, ref):
            assert torch.equal(a.indptr, b.indptr) and torch.equal(a.indices, b.indices) and torch.equal(a.self_idx, b.self_idx)
            assert a._t is not None
        def clone():
            o = OneBatchSubgraph(bt.adj_ens, [bt.feat_ens[0].clone()], bt.label, bt.size_subg_ens, bt.target_ens, [{}])
            o.tail_ens = bt.tail_ens
            return o
        m.prune_tail = False
        full = m.step(VALID, "running", clone())["preds"]
        m.prune_tail = True
        pruned = m.step(VALID, "running", clone())["preds"]
        np.testing.assert_allclose(pruned.cpu().numpy(), full.cpu().numpy(), rtol=1e-5, atol=1e-6)
        m.step(TRAIN, "running", clone())          # a training step through the pruned path


@pytest.mark.gpu
def test_backward_levels_square_form_without_the_sort_equals_the_sorted_one():
    """tail.build_backward_levels with ``targets_ascending`` (a collated batch's roots: one per subgraph, in order): the levels'
    square forms are built from the row lengths -- no argsort, no bincount (a host read-back) -- and equal the sorted
    construction entry by entry: row pointers, column ids, edge order, and the transposed form the attention backward walks."""
    from shadow_gnn_amd import tail
    _make, _A, n = _batch(24, 700, 16, 5, seed=11, deg=2)
    csr = _make().adj_ens[0]
    tgt = (torch.arange(24) * 700).to(DEV)
    a = tail.build_backward_levels(csr, tgt, max_levels=2, frac=0.6, targets_ascending=False)
    b = tail.build_backward_levels(csr, tgt, max_levels=2, frac=0.6, targets_ascending=True)
    assert len(a) == len(b) == 2
    assert [lv.rows_ascending for lv in a] == [False, True] and [lv.rows_ascending for lv in b] == [True, True]
    for la, lb in zip(a, b):
        (ca, oa), (cb, ob) = la.square, lb.square
        assert torch.equal(ca.indptr, cb.indptr) and torch.equal(ca.indices, cb.indices) and torch.equal(oa, ob)
        for x, y in zip(ca.transposed, cb.transposed):
            assert torch.equal(x, y)
        assert torch.equal(la.in_ids_full, lb.in_ids_full) and torch.equal(la.self_idx, lb.self_idx)
