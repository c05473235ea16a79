"""Pins oracle/layers_oracle.py against the reference's own layer / model outputs
(golden fixtures).  CPU only.  fp32 tolerance 1e-4 (absolute + relative), as
BASELINE.json's north_star states for floating point."""
import numpy as np
import torch

from oracle import layers_oracle as lo
from tests._golden_layers import LayerGolden, ModelGolden

TOL = dict(rtol=1e-4, atol=1e-4)


def _t(d):
    return {k: torch.tensor(v) for k, v in d.items()}


def test_layer_forward_and_gradients_match_reference():
    g = LayerGolden()
    assert len(g.cases) == 8
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


def test_model_step_matches_reference():
    g = ModelGolden()
    assert len(g.cases) == 4
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
