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


def _train_sampler_sections():
    """The DISTINCT (sampler section, self-edge rule, augmentations) combinations of the reference's training configurations:
    the `sampler` entries with phase 'train' as the files spell them; add_self_edge by the architecture (shaDow/utils.py:127-131:
    gcn / gat get self edges); the encodings the architecture's feature_augment asks the sampler for."""
    seen = {}
    for c in CONFIGS:
        arch = c["architecture"]
        aug = [] if arch.get("feature_augment", "none") in ("none", None) else sorted(str(arch["feature_augment"]).split("-"))
        self_e = arch["aggr"] in ("gcn", "gat", "gatscat")
        for s in c["sampler"]:
            if s.get("phase") != "train":
                continue
            key = json.dumps([{k: v for k, v in s.items() if k != "phase"}, self_e, aug], sort_keys=True)
            seen.setdefault(key, c["name"])
    return [(json.loads(k), name) for k, name in sorted(seen.items())]


_SECTIONS = _train_sampler_sections()


@pytest.mark.parametrize("sec,cfg_name", _SECTIONS, ids=[f"{n}:{s[0]['method']}" for s, n in _SECTIONS])
def test_reference_config_sampler_section_matches_oracle(sec, cfg_name):
    """The SAMPLER section of every training configuration the reference ships (every distinct combination of method, k /
    threshold / epsilon or depth / budget, self-edge rule and requested encodings): the HIP path -- PPR push for the table,
    then the sampler -- against the CPU oracle on a seeded graph, bit-exact (node sets, CSR, edge ids, encodings, fp32 scores);
    `ppr_st` (one configuration) is refused loudly."""
    from oracle import sampler_oracle as so
    from shadow_gnn_amd.ppr import ppr_approximate_device
    from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
    from shadow_gnn_amd.synthetic import make_graph_numpy
    from tests.test_sampler_gpu import _cmp_batch
    section, self_e, aug = sec
    method = section["method"]
    if method == "ppr_st":
        with pytest.raises(NotImplementedError):
            SamplerConfig.from_cpp_dict({"method": "ppr_st", "num_roots": "1", "k": "200", "threshold": "0.01"})
        return
    assert all(len(v) == 1 for k, v in section.items() if k != "method"), "one sampler per configuration (no ensembles in config_train)"
    indptr, indices = make_graph_numpy(20000, 14, seed=5)
    targets = np.random.default_rng(3).permutation(20000)[:192].astype(np.uint32)
    hs = HipSampler(indptr, indices, device=torch.device(DEV), seed=11)
    aug_s = tuple(a for a in aug if a in ("hops", "pprs", "drnls"))
    if method == "ppr":
        k, eps = int(section["k"][0]), float(section["epsilon"][0])
        thr = float(section.get("threshold", [0.0])[0])
        tab = so.ppr_approximate(indptr, indices, targets, k=k, alpha=0.85, epsilon=eps, num_threads=8)
        gl, gn, gs = ppr_approximate_device(hs, targets, k, 0.85, eps)
        assert np.array_equal(gl, tab.len)
        for i in range(targets.size):
            L = int(gl[i])
            assert np.array_equal(gn[i, :L], tab.neigh[i, :L]) and np.array_equal(gs[i, :L].view(np.uint32), tab.score[i, :L].view(np.uint32)), i
        hs.set_ppr(targets, gl, gn, gs)
        hs.shuffle_targets(targets)
        b = hs.sample(SamplerConfig(method="ppr", k=k, threshold=thr, add_self_edge=self_e, aug=aug_s), targets.size)
        ref = so.sample_batch(indptr, indices, targets, method="ppr", k=k, threshold=thr, add_self_edge=self_e, aug=aug_s, ppr=tab)
    else:
        assert method == "khop"
        depth, budget = int(section["depth"][0]), int(section["budget"][0])
        hs.shuffle_targets(targets)
        b = hs.sample(SamplerConfig(method="khop", depth=depth, budget=budget, add_self_edge=self_e, aug=aug_s), targets.size)
        ref = so.sample_batch(indptr, indices, targets, method="khop", depth=depth, budget=budget, add_self_edge=self_e, aug=aug_s, seed=11)
    _cmp_batch(ref, b, aug_s, (cfg_name, method))
