"""Fuzz of the row-sparse top-layer backward passes (GraphSAGE: ops._SageDense._sparse_top_backward / _compact_dz_backward;
GAT: ops_gat._GatTail._rows_backward over tail.build_backward_levels) against the dense passes on random ragged
block-diagonal batches: subgraphs of 1 .. 60 nodes, roots with and without self edges, roots without any neighbour, sparse and
dense blocks, 2-4 layers, dropout / drop-edge on -- loss, predictions and every parameter gradient must agree.
    python scripts/fuzz_sparse_top.py [seed] [trials]"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
DEV = "cuda:0"


def run(seed: int, trials: int, verbose: bool = True):
    from shadow_gnn_amd import ops, ops_gat
    from shadow_gnn_amd.minibatch import TRAIN, OneBatchSubgraph
    from shadow_gnn_amd.models import DeepGNN
    rng = np.random.default_rng(seed)
    failures = []
    saved = (ops.SPARSE_TOP_BWD, ops.SPARSE_TOP_BWD_MIN_ROWS, ops.GEMM_SPLIT_MIN_ROWS, ops.AMAX_HANDOVER_ROWS, ops.BACKWARD_LEVELS_FRAC)
    ops.SPARSE_TOP_BWD_MIN_ROWS = 1
    ops.BACKWARD_LEVELS_FRAC = 2.0                   # (GAT: keep both levels whatever share of these tiny batches they cover)
    ops.GEMM_SPLIT_MIN_ROWS = 1                      # (tiny batches through the one-call entries, as the golden tests do)
    used = 0
    try:
        for trial in range(trials):
            aggr = str(rng.choice(["sage", "gat"]))
            heads = int(rng.choice([1, 2, 4])) if aggr == "gat" else 1
            dim = int(rng.choice([32, 64, 96])) if aggr == "sage" else int(rng.choice([32, 64]))      # (GAT: head slices of 8 .. 64 floats)
            L = int(rng.integers(2, 5)) if aggr == "sage" else int(rng.integers(1, 4))
            p_drop, p_edge = float(rng.choice([0.0, 0.3])), float(rng.choice([0.0, 0.2]))
            B = int(rng.choice([1, 4, 17])); F0 = int(rng.choice([8, 20, 32])); C = int(rng.choice([2, 5]))
            sizes = rng.integers(1, 60, B)
            blocks = []
            for s_ in sizes:
                a = (rng.random((s_, s_)) < rng.choice([0.03, 0.15, 0.5])).astype(np.float32)
                a = np.maximum(a, a.T)
                np.fill_diagonal(a, 1.0 if (aggr == "gat" or rng.random() < 0.4) else 0.0)
                blocks.append(sp.csr_matrix(a))
            A = sp.block_diag(blocks, format="csr"); A.sort_indices(); A.eliminate_zeros()
            n = A.shape[0]
            off = np.concatenate([[0], np.cumsum(sizes)])
            target = (off[:-1] + rng.integers(0, sizes)).astype(np.int64)
            arch = dict(num_layers=L, num_cls_layers=1, heads=heads, dim=dim, act=str(rng.choice(["relu", "elu"])), layer_norm="norm_feat",
                        feature_augment_ops="sum", aggr=aggr, residue="none", pooling="center", loss="softmax")
            ctx = (trial, aggr, L, dim, heads, B, n, p_drop, p_edge)
            X = torch.randn(n, F0, generator=torch.Generator().manual_seed(trial))
            labels = torch.as_tensor(rng.integers(0, C, B))
            res = []
            try:
                for sparse in (False, True):
                    ops.SPARSE_TOP_BWD = sparse
                    torch.manual_seed(1000 + trial)
                    m = DeepGNN(F0, F0, C, 0, arch, [], 1, dict(lr=0.01, dropout=p_drop, dropedge=p_edge), "node").to(DEV)
                    with torch.no_grad():
                        for q in m.parameters():
                            q.add_(0.1 * torch.randn_like(q))
                    m.optimizer = torch.optim.SGD(m.parameters(), lr=0.0)
                    csr = ops.DeviceCSR(torch.from_numpy(A.indptr.astype(np.int32)).to(DEV), torch.from_numpy(A.indices.astype(np.int32)).to(DEV),
                                        subg_off=torch.from_numpy(off.astype(np.int32)).to(DEV),
                                        subg_edge_off=torch.from_numpy(A.indptr[off].astype(np.int32)).to(DEV), max_subg_nodes=int(sizes.max()))
                    bt = OneBatchSubgraph([csr], [X.to(DEV)], labels.to(DEV), torch.as_tensor(sizes.astype(np.int64)).to(DEV).unsqueeze(0),
                                          [torch.as_tensor(target).to(DEV)], [{}])
                    c0 = ops._SageDense.sparse_top_calls + ops_gat._GatTail.sparse_top_calls
                    torch.manual_seed(2000 + trial)             # (dropout seeds / drop-edge draws)
                    ret = m.step(TRAIN, "running", bt)
                    torch.cuda.synchronize()
                    res.append((float(ret["loss"]), ret["preds"].detach().clone(), {k: q.grad.detach().clone() for k, q in m.named_parameters()},
                                ops._SageDense.sparse_top_calls + ops_gat._GatTail.sparse_top_calls - c0))
                (l0, p0, g0, k0), (l1, p1, g1, k1) = res
                assert k0 == 0
                used += 1 if k1 else 0
                assert abs(l0 - l1) < 1e-5, ("loss", l0, l1)
                assert torch.equal(p0, p1), "predictions"
                for k in g0:
                    scale = float(g0[k].abs().max())
                    err = float((g1[k] - g0[k]).abs().max())
                    assert err <= 5e-5 * scale + 1e-8, (k, err, scale)
            except Exception as ex:                              # noqa: BLE001
                failures.append((ctx, f"{type(ex).__name__}: {ex}"[:400]))
                if verbose:
                    print("FAIL", ctx, failures[-1][1])
    finally:
        ops.SPARSE_TOP_BWD, ops.SPARSE_TOP_BWD_MIN_ROWS, ops.GEMM_SPLIT_MIN_ROWS, ops.AMAX_HANDOVER_ROWS, ops.BACKWARD_LEVELS_FRAC = saved
    if verbose:
        print(f"{trials} trials, {used} took a row-sparse pass, {len(failures)} failures")
    return failures, used


if __name__ == "__main__":
    f, u = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    sys.exit(1 if f else 0)
