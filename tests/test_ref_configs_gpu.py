"""Every training configuration the reference ships (config_train/*/*/*.yml; the architecture / hyper-parameter
sections are kept as data in tests/golden/ref_config_archs.json) either builds and trains on the HIP path, or is
refused loudly for the documented out-of-scope feature (sort pooling, needs PyG) -- never silently wrong."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = json.load(open(os.path.join(HERE, "golden", "ref_config_archs.json")))
DIM_AUG = {"hops": 7, "pprs": 1, "drnls": 26}            # minibatch.py:246-248


def _batch(B, n_per, F0, C, aug, multilabel):
    from shadow_gnn_amd import ops
    from shadow_gnn_amd.minibatch import OneBatchSubgraph
    rng = np.random.default_rng(0)
    blocks = []
    for b in range(B):
        a = (rng.random((n_per, n_per)) < 0.08).astype(np.float32)
        a = np.maximum(a, a.T); np.fill_diagonal(a, 1.0)
        blocks.append(sp.csr_matrix(a))
    A = sp.block_diag(blocks, format="csr"); A.sort_indices()
    n = A.shape[0]
    csr = ops.DeviceCSR(torch.from_numpy(A.indptr.astype(np.int32)).to(DEV), torch.from_numpy(A.indices.astype(np.int32)).to(DEV))
    g = torch.Generator(device=DEV).manual_seed(1)
    feat = torch.randn(n, F0, device=DEV, generator=g)
    fa = {}
    for a_ in aug:
        codes = torch.randint(0, DIM_AUG[a_], (n,), device=DEV, generator=g)
        fa[a_] = torch.nn.functional.one_hot(codes, DIM_AUG[a_]).float()
    label = (torch.randint(0, 2, (B, C), device=DEV, generator=g).float() if multilabel
             else torch.randint(0, C, (B,), device=DEV, generator=g))
    tgt = (torch.arange(B) * n_per).to(DEV)
    return OneBatchSubgraph([csr], [feat], label, torch.full((1, B), n_per, dtype=torch.int64, device=DEV), [tgt], [fa])


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c["name"] for c in CONFIGS])
def test_reference_config_builds_and_trains(cfg):
    from shadow_gnn_amd.minibatch import TRAIN, VALID
    from shadow_gnn_amd.models import DeepGNN
    arch = dict(cfg["architecture"])
    aug = [] if arch.get("feature_augment", "none") in ("none", None) else str(arch["feature_augment"]).split("-")
    unsupported = str(arch.get("pooling", "center")).startswith("sort")
    F0, C, B, n_per = 24, 6, 12, 40
    aug_feat = [(a, DIM_AUG[a]) for a in aug]
    dim_in = F0
    torch.manual_seed(0)

    def build():
        return DeepGNN(dim_in, dim_in, C, 0, arch, aug_feat, 1,
                       dict(lr=float(cfg["lr"] or 0.001), dropout=float(cfg["dropout"] or 0.0), dropedge=float(cfg["dropedge"] or 0.0)),
                       "node").to(DEV)
    if unsupported:
        with pytest.raises(NotImplementedError):
            m = build()
            m.step(TRAIN, "running", _batch(B, n_per, F0, C, aug, arch.get("loss") == "sigmoid"))
        return
    m = build()
    losses = []
    for _ in range(3):
        ret = m.step(TRAIN, "running", _batch(B, n_per, F0, C, aug, arch.get("loss") == "sigmoid"))
        losses.append(float(ret["loss"].detach()))
    assert all(np.isfinite(losses)), losses
    assert all(p_.grad is None or torch.isfinite(p_.grad).all() for p_ in m.parameters())
    ev = m.step(VALID, "running", _batch(B, n_per, F0, C, aug, arch.get("loss") == "sigmoid"))
    assert ev["preds"].shape == (B, C) and torch.isfinite(ev["preds"]).all()
