"""Pins oracle/layers_oracle.py against the reference's own layer / model outputs
(golden fixtures).  CPU only.  fp32 tolerance 1e-4 (absolute + relative), as
BASELINE.json's north_star states for floating point."""
import numpy as np
import pytest
import torch

from oracle import layers_oracle as lo
from tests._golden_layers import LayerGolden, ModelGolden

TOL = dict(rtol=1e-4, atol=1e-4)


def _t(d):
    return {k: torch.tensor(v) for k, v in d.items()}


@pytest.mark.parametrize("fname,ncases", [("layers_fwd_bwd.npz", 8), ("layers_prelu.npz", 5)])
def test_layer_forward_and_gradients_match_reference(fname, ncases):
    g = LayerGolden(fname)
    assert len(g.cases) == ncases
    for case in g.cases:
        ci = case["idx"]
        p = {k: v.clone().requires_grad_(True) for k, v in _t(g.params(ci)).items()}
        X = torch.tensor(g.get(ci, "X"), requires_grad=True)
        out = lo.layer_forward(case["layer"], p, X, g.get(ci, "indptr"), g.get(ci, "indices"),
                               case["act"], heads=case.get("mulhead", 1))
        np.testing.assert_allclose(out.detach().numpy(), g.get(ci, "out"), err_msg=str(case), **TOL)
        (out * torch.tensor(g.get(ci, "wout"))).sum().backward()
        np.testing.assert_allclose(X.grad.numpy(), g.get(ci, "dX"), err_msg=str(case), **TOL)
        for k, gr in g.grads(ci).items():
            np.testing.assert_allclose(p[k].grad.numpy(), gr, err_msg=f"{case} {k}", **TOL)
        # layers 1..L-1 receive the already normalised adjacency: same result as normalising again
        p2 = _t(g.params(ci, "p2"))
        out2 = lo.layer_forward(case["layer"], p2, torch.tensor(g.get(ci, "out")), g.get(ci, "indptr"),
                                g.get(ci, "indices"), case["act"], heads=case.get("mulhead", 1))
        np.testing.assert_allclose(out2.detach().numpy(), g.get(ci, "out2"), err_msg=str(case), **TOL)


@pytest.mark.parametrize("fname,ncases", [("models_step.npz", 4), ("models_prelu.npz", 2)])
def test_model_step_matches_reference(fname, ncases):
    g = ModelGolden(fname)
    assert len(g.cases) == ncases
    for case in g.cases:
        ci = case["idx"]
        p = {k: v.clone().requires_grad_(True) for k, v in _t(g.group(ci, "p")).items()}
        hop1hot = None
        if case["aug"]:
            hop1hot = torch.tensor(lo.hop2onehot(g.get(ci, "hop"), 7))
            np.testing.assert_array_equal(hop1hot.numpy(), g.get(ci, "hop1hot"))
        preds, emb = lo.model_forward(p, case["arch"], torch.tensor(g.get(ci, "X")), g.get(ci, "indptr"),
                                      g.get(ci, "indices"), g.get(ci, "sizes"), g.get(ci, "target"), hop1hot)
        np.testing.assert_allclose(preds.detach().numpy(), g.get(ci, "preds"), err_msg=str(case["arch"]), **TOL)
        np.testing.assert_allclose(emb.detach().numpy(), g.get(ci, "emb"), **TOL)
        loss = lo.model_loss(preds, g.get(ci, "labels"))
        assert abs(float(loss) - float(g.get(ci, "loss"))) < 1e-4
        loss.backward()
        params = [v for v in p.values() if v.grad is not None]
        gn = torch.nn.utils.clip_grad_norm_(params, 5)
        assert abs(float(gn) - float(g.get(ci, "gnorm"))) < 1e-3 * max(1.0, float(gn))
        for k, gr in g.group(ci, "g").items():
            np.testing.assert_allclose(p[k].grad.numpy(), gr, err_msg=f"{case['arch']['aggr']} {k}", rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("kind", ["sage", "gcn"])
def test_cpu_train_step_port_matches_layer_oracle(kind):
    """oracle/cpu_train_step.py (the sparse-matrix CPU model bench.py times as ``cpu_baseline_train_step``)
    computes what the golden-pinned dense layer oracle computes."""
    from oracle import cpu_train_step as cts
    from oracle import layers_oracle as lo
    rng = np.random.default_rng(5)
    blocks, n_per, B = [], 12, 5
    import scipy.sparse as sp
    for b in range(B):
        a = (rng.random((n_per, n_per)) < 0.3).astype(np.float32)
        a = np.maximum(a, a.T); np.fill_diagonal(a, 1.0)
        blocks.append(sp.csr_matrix(a))
    A = sp.block_diag(blocks, format="csr"); A.sort_indices()
    n, F0, dim, C, L = A.shape[0], 7, 16, 4, 3
    torch.manual_seed(1)
    X = torch.randn(n, F0)
    target = np.arange(B) * n_per
    m = cts.CpuModel(kind, L, F0, dim, C, "relu", dropout=0.0).eval()
    with torch.no_grad():
        for q in m.parameters():
            q.add_(0.1 * torch.randn_like(q))          # (scale / offset away from their 1 / 0 initial values)
    p = {}
    for l, layer in enumerate(m.layers):
        pre = f"conv_layers.0.{l}."
        if kind == "gcn":
            p[pre + "f_lin.weight"], p[pre + "f_lin.bias"] = layer.lins[0].weight, layer.lins[0].bias
        else:
            p[pre + "f_lin_self.weight"], p[pre + "f_lin_self.bias"] = layer.lins[0].weight, layer.lins[0].bias
            p[pre + "f_lin_neigh.weight"], p[pre + "f_lin_neigh.bias"] = layer.lins[1].weight, layer.lins[1].bias
        p[pre + "scale"], p[pre + "offset"] = layer.scale, layer.offset
    p["classifier.0.f_lin.weight"], p["classifier.0.f_lin.bias"] = m.cls.weight, m.cls.bias
    p["classifier.0.scale"], p["classifier.0.offset"] = m.cls_scale.unsqueeze(0), m.cls_offset.unsqueeze(0)
    arch = dict(aggr=kind, num_layers=L, heads=1, act="relu", residue="none", pooling="center")
    ref, _ = lo.model_forward(p, arch, X, A.indptr, A.indices, [n_per] * B, target)
    adj = cts.norm_adj(A.indptr, A.indices, kind, 0.0, torch.Generator().manual_seed(0))
    got = m(X, adj, torch.as_tensor(target))
    np.testing.assert_allclose(got.detach().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    # and a whole optimisation step runs
    steps, sec, warm = cts.time_train_steps(A.indptr, A.indices, X, torch.as_tensor(target), torch.arange(B) % C, kind, L, dim, C,
                                            "relu", 0.3, 0.1, 0.01, threads=2, budget_s=5.0, max_steps=1)
    assert steps == 1 and sec > 0


@pytest.mark.parametrize("fname", ["models_step.npz", "models_prelu.npz"])
def test_sparse_model_oracle_matches_reference_golden(fname):
    """oracle/model_oracle_sparse.py (the fp64 edge-list checker of the benchmark-scale GPU parity tests) reproduces
    the reference's own DeepGNN.step vectors: predictions, embeddings, loss and every parameter gradient."""
    from oracle import model_oracle_sparse as mos
    g = ModelGolden(fname)
    for case in g.cases:
        ci = case["idx"]
        p = {k: v.double().requires_grad_(True) for k, v in _t(g.group(ci, "p")).items()}
        hop1hot = torch.tensor(lo.hop2onehot(g.get(ci, "hop"), 7)) if case["aug"] else None
        preds, emb = mos.model_forward(p, case["arch"], torch.tensor(g.get(ci, "X")), g.get(ci, "indptr"),
                                       g.get(ci, "indices"), g.get(ci, "sizes"), g.get(ci, "target"), hop1hot)
        assert preds.dtype == torch.float64
        np.testing.assert_allclose(preds.detach().numpy(), g.get(ci, "preds"), err_msg=str(case["arch"]), **TOL)
        np.testing.assert_allclose(emb.detach().numpy(), g.get(ci, "emb"), **TOL)
        loss = lo.model_loss(preds, g.get(ci, "labels"))
        assert abs(float(loss) - float(g.get(ci, "loss"))) < 1e-4
        loss.backward()
        for k, gr in g.group(ci, "g").items():
            np.testing.assert_allclose(p[k].grad.numpy(), gr, err_msg=f"{case['arch']['aggr']} {k}", rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("kind,heads", [("sage", 1), ("gcn", 1), ("gat", 2)])
def test_sparse_model_oracle_matches_dense_oracle(kind, heads):
    """Same arithmetic as the golden-pinned dense oracle on a ragged random batch with isolated rows (fp32 both)."""
    import scipy.sparse as sp
    from oracle import model_oracle_sparse as mos
    rng = np.random.default_rng(11)
    sizes = rng.integers(1, 30, 9)
    blocks = []
    for s_ in sizes:
        a = (rng.random((s_, s_)) < 0.2).astype(np.float32)
        a = np.maximum(a, a.T)
        if kind != "sage":
            np.fill_diagonal(a, 1.0)
        blocks.append(sp.csr_matrix(a))
    A = sp.block_diag(blocks, format="csr"); A.sort_indices()
    n, F0, dim, C, L = A.shape[0], 9, 8 * heads, 4, 3
    off = np.concatenate([[0], np.cumsum(sizes)])
    target = off[:-1] + rng.integers(0, sizes)
    torch.manual_seed(3)
    p = {}
    for l in range(L):
        pre, fi = f"conv_layers.0.{l}.", (F0 if l == 0 else dim)
        if kind == "gcn":
            p[pre + "f_lin.weight"], p[pre + "f_lin.bias"] = torch.randn(dim, fi) * 0.3, torch.randn(dim) * 0.1
            nb = (1, dim)
        elif kind == "sage":
            for nm in ("f_lin_self", "f_lin_neigh"):
                p[pre + nm + ".weight"], p[pre + nm + ".bias"] = torch.randn(dim, fi) * 0.3, torch.randn(dim) * 0.1
            nb = (2, dim)
        else:
            for j in range(2):
                p[pre + f"f_lin.{j}.weight"], p[pre + f"f_lin.{j}.bias"] = torch.randn(dim, fi) * 0.3, torch.randn(dim) * 0.1
            p[pre + "attention"] = torch.randn(2, heads, dim // heads) * 0.5
            nb = (2, heads, dim // heads)
        p[pre + "scale"], p[pre + "offset"] = torch.rand(nb) + 0.5, torch.randn(nb) * 0.1
    p["res_pool_layers.0.nn.1.weight"], p["res_pool_layers.0.nn.1.bias"] = torch.randn(dim, 2 * dim) * 0.3, torch.randn(dim) * 0.1
    p["res_pool_layers.0.scale"], p["res_pool_layers.0.offset"] = torch.rand(dim) + 0.5, torch.randn(dim) * 0.1
    p["classifier.0.f_lin.weight"], p["classifier.0.f_lin.bias"] = torch.randn(C, dim) * 0.3, torch.randn(C) * 0.1
    p["classifier.0.scale"], p["classifier.0.offset"] = torch.rand(1, C) + 0.5, torch.randn(1, C) * 0.1
    arch = dict(aggr=kind, num_layers=L, heads=heads, act="elu", residue="max", pooling="mean")
    X = torch.randn(n, F0)
    labels = rng.integers(0, C, sizes.size)
    pd = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ps = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref, ref_emb = lo.model_forward(pd, arch, X, A.indptr, A.indices, sizes, target)
    got, got_emb = mos.model_forward(ps, arch, X, A.indptr, A.indices, sizes, target, dtype=torch.float32)
    np.testing.assert_allclose(got.detach().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(got_emb.detach().numpy(), ref_emb.detach().numpy(), rtol=1e-4, atol=1e-5)
    lo.model_loss(ref, labels).backward(); lo.model_loss(got, labels).backward()
    for k in p:
        np.testing.assert_allclose(ps[k].grad.numpy(), pd[k].grad.numpy(), rtol=1e-3, atol=1e-5, err_msg=k)
