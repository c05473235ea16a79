#!/usr/bin/env python3
"""Generate layer / model golden vectors under tests/golden/ from the REFERENCE
python code (shaDow/layers.py, shaDow/models.py), imported from /root/reference
in this container.

TEST INFRASTRUCTURE ONLY.  Outputs are data (.npz): inputs (block-diagonal batch
CSR, features, sizes, targets, parameter tensors) and the reference's outputs /
gradients.  No reference source text is stored.

Third-party modules the reference imports but this image lacks are handled as
SURVEY.md section 8(c) describes:
  * torch_scatter.scatter (unpinned version; README:164-166) -- only
    reduce in {"sum","max"} on 1-D fp32 is used (layers.py:572,573,578,
    graph_utils.py:64).  Its published semantics are restated below with
    Tensor.scatter_reduce (out[idx[i]] = reduce(src[i]), size = max(idx)+1).
  * torch_geometric / ogb -- never called on this path; stubbed to raise.

    python oracle/gen_golden_layers.py      # (re)writes tests/golden/layers_*.npz
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SHADOW_REFERENCE_ROOT", "/root/reference")

import numpy as np
import scipy.sparse as sp
import torch


def _install_stubs():
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
        assert src.dim() == 1 and index.dim() == 1 and reduce in ("sum", "max")
        size = int(index.max()) + 1 if dim_size is None else dim_size
        init = torch.zeros(size, dtype=src.dtype, device=src.device)
        return init.scatter_reduce(0, index, src, reduce={"sum": "sum", "max": "amax"}[reduce], include_self=False)
    ts.scatter = scatter
    sys.modules["torch_scatter"] = ts

    def _raise(*a, **k):
        raise RuntimeError("stubbed third-party function called on the golden path")
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    tgnn.global_sort_pool = _raise
    tgu = types.ModuleType("torch_geometric.utils")
    tgu.negative_sampling = tgu.add_self_loops = tgu.to_undirected = _raise
    tg.nn, tg.utils = tgnn, tgu
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tgnn, "torch_geometric.utils": tgu})
    for name in ("ogb", "ogb.nodeproppred", "ogb.linkproppred"):
        m = types.ModuleType(name)
        m.Evaluator = _raise
        sys.modules[name] = m


def import_reference():
    _install_stubs()
    os.chdir(REF)                                   # shaDow/globals.py reads CONFIG_TEMPLATE.yml from cwd
    sys.argv = ["x", "--dataset", "arxiv", "--gpu", "-1"]
    sys.path.insert(0, os.path.join(HERE, "_ref"))              # the reference's own C++ sampler module (oracle/build_ref.sh)
    sys.path.insert(0, os.path.join(REF, "para_graph_sampler"))
    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    import shaDow.layers as L
    import shaDow.models as M
    import shaDow.minibatch as MB
    return L, M, MB


def make_batch(seed, P=6, n_graph=120, avg_deg=6, self_edge=False, feat=12):
    """A small block-diagonal batch sampled by the (pinned) CPU oracle."""
    sys.path.insert(0, ROOT)
    from oracle import sampler_oracle as so
    from oracle.gen_golden import make_graph
    indptr, indices = make_graph(n_graph, avg_deg, seed=seed)
    rng = np.random.default_rng(seed + 1)
    roots = rng.choice(n_graph, P, replace=False).astype(np.uint32)
    b = so.sample_batch(indptr, indices, roots, method="khop", depth=2, budget=3, add_self_edge=self_edge,
                        aug=("hops",), seed=seed)
    X = rng.standard_normal((b.node.size, feat)).astype(np.float32)
    return b, X


def csr_of(b):
    n = b.node.size
    return sp.csr_matrix((np.ones(b.indices.size, dtype=np.float32), b.indices.astype(np.int64),
                          b.indptr.astype(np.int64)), shape=(n, n))


def tnp(t):
    return t.detach().cpu().numpy()


LAYER_CASES = lambda L: (
    ("gcn", L.GCN, True, dict(act="elu")), ("gcn", L.GCN, True, dict(act="relu")),
    ("sage", L.GraphSAGE, False, dict(act="elu")), ("sage", L.GraphSAGE, False, dict(act="relu")),
    ("sage", L.GraphSAGE, True, dict(act="tanh")),
    ("gat", L.GAT, True, dict(act="elu", mulhead=4)), ("gat", L.GAT, True, dict(act="relu", mulhead=2)),
    ("gat", L.GAT, False, dict(act="elu", mulhead=1)),
)
# learnable activations (nn.PReLU: one slope per layer; "prelu+": one per output channel; layers.py:26-39)
LAYER_CASES_PRELU = lambda L: (
    ("gcn", L.GCN, True, dict(act="prelu")), ("sage", L.GraphSAGE, False, dict(act="prelu")),
    ("sage", L.GraphSAGE, True, dict(act="prelu+")), ("gat", L.GAT, True, dict(act="prelu", mulhead=2)),
    ("gat", L.GAT, False, dict(act="prelu+", mulhead=4)),
)


def gen_layers(L, case_list=None, fname="layers_fwd_bwd.npz", seed=0):
    store = {}
    cases = []
    torch.manual_seed(seed)
    ci = 0
    for (name, cls, self_edge, kw) in (case_list or LAYER_CASES(L)):
        b, X = make_batch(seed=10 + ci, self_edge=self_edge, feat=12)
        dim_in, dim_out = 12, 16
        layer = cls(dim_in, dim_out, dropout=0.0, norm="norm_feat", **kw)
        with torch.no_grad():
            for p in layer.parameters():          # non-trivial scale/offset/bias
                p.add_(0.3 * torch.randn_like(p))
        layer.train()
        x = torch.tensor(X, requires_grad=True)
        sizes = torch.tensor(b.subg_nodes.astype(np.int64))
        out, adj_norm, flag, de = layer((x, csr_of(b), False, 0.0), sizes)
        assert flag is True and de == 0.0
        w = torch.tensor(np.random.default_rng(ci).standard_normal(out.shape).astype(np.float32))
        (out * w).sum().backward()
        pre = f"c{ci}"
        store[f"{pre}_indptr"] = b.indptr; store[f"{pre}_indices"] = b.indices
        store[f"{pre}_sizes"] = b.subg_nodes; store[f"{pre}_X"] = X; store[f"{pre}_wout"] = tnp(w)
        store[f"{pre}_out"] = tnp(out); store[f"{pre}_dX"] = tnp(x.grad)
        for k, v in layer.state_dict().items():
            store[f"{pre}_p_{k}"] = tnp(v)
        for k, p in layer.named_parameters():
            store[f"{pre}_g_{k}"] = tnp(p.grad)
        # second application with the threaded (already normalised) adjacency, like layers 1..L-1
        x2 = torch.tensor(tnp(out))
        out2 = cls(dim_out, dim_out, dropout=0.0, norm="norm_feat", **kw)
        out2.load_state_dict({k: (v if v.shape == out2.state_dict()[k].shape else out2.state_dict()[k])
                              for k, v in layer.state_dict().items()}, strict=False)
        o2, _, _, _ = out2((x2, adj_norm, True, 0.0), sizes)
        store[f"{pre}_out2"] = tnp(o2)
        for k, v in out2.state_dict().items():
            store[f"{pre}_p2_{k}"] = tnp(v)
        cases.append(dict(idx=ci, layer=name, dim_in=dim_in, dim_out=dim_out, **kw))
        ci += 1
    import json
    store["cases"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", fname)
    np.savez_compressed(path, **store)
    print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path)/1024:.1f} KiB")


MODEL_CASES = (
    ("sage", False, 1, "none", "center", "elu", True, 3),
    ("gcn", True, 1, "none", "center", "elu", True, 3),
    ("gat", True, 4, "none", "center", "elu", False, 2),
    ("sage", False, 1, "max", "mean", "relu", False, 3),
)
MODEL_CASES_PRELU = (
    ("gat", True, 4, "max", "max", "prelu", False, 2),       # the leaderboard GAT read-out with the ResPool PReLU
    ("sage", False, 1, "none", "center", "prelu", True, 3),
)


def gen_models(L, M, MB, case_list=MODEL_CASES, fname="models_step.npz", seed_base=100):
    import json
    store, cases = {}, []
    ci = 0
    for (aggr, self_edge, heads, residue, pooling, act, aug, nl) in case_list:
        torch.manual_seed(seed_base + ci)
        b, X = make_batch(seed=50 + ci, P=8, self_edge=self_edge, feat=10)
        num_classes = 5
        arch = dict(num_layers=nl, num_cls_layers=1, heads=heads, branch_sharing=False, dim=16, act=act,
                    layer_norm="norm_feat", feature_augment_ops="sum", aggr=aggr, residue=residue,
                    pooling=pooling, loss="softmax", ensemble_act="relu")
        tp = dict(dropout=0.0, dropedge=0.0, lr=0.01, ensemble_dropout="none")
        aug_feat = [("hops", 7)] if aug else []
        model = M.DeepGNN(10, 10, num_classes, 0, arch, aug_feat, 1, tp, "node")
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.2 * torch.randn_like(p))
        rng = np.random.default_rng(ci)
        labels = rng.integers(0, num_classes, b.subg_nodes.size)
        feat_aug = {}
        if aug:
            from graph_engine.frontend.graph import EntityEncoding
            enc = EntityEncoding(hop=b.hop.astype(np.int64), validate=False)
            feat_aug["hops"] = enc.hop2onehot_vec(7, return_type="tensor").type(torch.float32)
            store[f"m{ci}_hop"] = b.hop
            store[f"m{ci}_hop1hot"] = tnp(feat_aug["hops"])
        batch = MB.OneBatchSubgraph([csr_of(b)], [torch.tensor(X.copy())], torch.tensor(labels),
                                    torch.tensor(b.subg_nodes.astype(np.int64)).unsqueeze(0),
                                    [b.target.astype(np.int64)], [feat_aug])
        state0 = {k: v.clone() for k, v in model.state_dict().items()}
        # forward + backward exactly as DeepGNN.step does it (models.py:217-224), keeping the grads
        model.train()
        model.optimizer.zero_grad()
        preds, emb = model(0, dropedge=0.0, **batch.to_dict({"feat_ens", "adj_ens", "target_ens", "size_subg_ens", "feat_aug_ens"}))
        lab1h = torch.nn.functional.one_hot(torch.tensor(labels), num_classes=num_classes)
        loss = model._loss(preds, lab1h)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
        grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        model.optimizer.step()
        pre = f"m{ci}"
        store[f"{pre}_indptr"] = b.indptr; store[f"{pre}_indices"] = b.indices
        store[f"{pre}_sizes"] = b.subg_nodes; store[f"{pre}_target"] = b.target; store[f"{pre}_X"] = X
        store[f"{pre}_labels"] = labels
        store[f"{pre}_preds"] = tnp(preds); store[f"{pre}_loss"] = np.array(float(loss)); store[f"{pre}_gnorm"] = np.array(float(gn))
        store[f"{pre}_emb"] = tnp(emb[0])
        for k, v in state0.items():
            store[f"{pre}_p_{k}"] = tnp(v)
        for k, g in grads.items():
            store[f"{pre}_g_{k}"] = tnp(g)
        for k, v in model.state_dict().items():
            store[f"{pre}_q_{k}"] = tnp(v)             # parameters after one Adam step
        cases.append(dict(idx=ci, arch=arch, train_params=tp, aug=aug, num_classes=num_classes, dim_feat=10))
        ci += 1
    store["cases"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", fname)
    np.savez_compressed(path, **store)
    print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path)/1024:.1f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(4)
    prelu_only = "--prelu-only" in sys.argv          # (import_reference() replaces sys.argv)
    L, M, MB = import_reference()
    if not prelu_only:
        gen_layers(L)
        gen_models(L, M, MB)
    gen_layers(L, LAYER_CASES_PRELU(L), "layers_prelu.npz", seed=7)
    gen_models(L, M, MB, MODEL_CASES_PRELU, "models_prelu.npz", seed_base=300)
