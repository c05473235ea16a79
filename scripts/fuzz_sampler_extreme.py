"""One-off extreme-structure fuzz of the HIP sampler against the oracle: tiny graphs, empty graphs, complete
graphs, stars, paths, duplicate roots; k-hop / nodeIID / PPR, every flag.  Prints the first mismatch."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import sampler_oracle as so
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from tests.test_sampler_gpu import _cmp_batch
so.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
def csr(n, a, b):
    if len(a):
        key = np.unique(np.asarray(a, dtype=np.int64) * n + np.asarray(b, dtype=np.int64))
        rows, cols = key // n, (key % n).astype(np.uint32)
    else:
        rows, cols = np.zeros(0, np.int64), np.zeros(0, np.uint32)
    ip = np.zeros(n + 1, dtype=np.int64); np.add.at(ip, rows + 1, 1)
    return np.cumsum(ip).astype(np.uint32), cols
def graph(kind, n):
    if kind == "empty": return csr(n, [], [])
    if kind == "complete": a, b = np.meshgrid(np.arange(n), np.arange(n)); a, b = a.ravel(), b.ravel(); k = a != b; return csr(n, a[k], b[k])
    if kind == "complete_loops": a, b = np.meshgrid(np.arange(n), np.arange(n)); return csr(n, a.ravel(), b.ravel())
    if kind == "star": a = np.zeros(n - 1, int); b = np.arange(1, n); return csr(n, np.concatenate([a, b]), np.concatenate([b, a]))
    if kind == "path": a = np.arange(n - 1); return csr(n, np.concatenate([a, a + 1]), np.concatenate([a + 1, a]))
    if kind == "dirpath": a = np.arange(n - 1); return csr(n, a, a + 1)
    m = int(n * rng.choice([0.3, 1.5, 4])); return csr(n, rng.integers(0, n, m), rng.integers(0, n, m))
bad = 0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
for trial in range(T):
    kind = str(rng.choice(["empty", "complete", "complete_loops", "star", "path", "dirpath", "random", "random"]))
    n = int(rng.choice([1, 2, 3, 5, 8, 17, 64, 65, 130, 700])) if kind not in ("complete", "complete_loops") else int(rng.choice([1, 2, 5, 33, 64, 90]))
    if kind in ("star", "path", "dirpath") and n < 2: n = 2
    indptr, indices = graph(kind, n)
    num_roots = int(rng.choice([1, 1, 2]))
    P = int(rng.choice([1, 2, 7, 40]))
    roots = rng.integers(0, n, P * num_roots).astype(np.uint32)
    if rng.random() < 0.3: roots[:] = roots[0]                     # duplicate roots
    method = str(rng.choice(["khop", "khop", "nodeIID", "ppr"]))
    kw = dict(method=method, num_roots=num_roots, add_self_edge=bool(rng.random() < 0.5), include_target_conn=bool(rng.random() < 0.5),
              compat_overread=bool(rng.random() < 0.2))
    if method == "khop": kw.update(depth=int(rng.integers(0, 5)), budget=int(rng.choice([-1, 0, 1, 2, 20, 1000])))
    aug = tuple(x for x in ("hops", "drnls", "pprs") if rng.random() < 0.4 and (x != "drnls" or num_roots == 2))
    seed = int(rng.integers(0, 2 ** 31))
    try:
        hs = HipSampler(indptr, indices, device=torch.device("cuda:0"), seed=seed)
        extra = {}
        if method == "ppr":
            k = int(rng.choice([1, 3, 50])); thr = float(rng.choice([0.0, 0.01, 0.5]))
            uniq = np.unique(roots)
            tab = so.ppr_approximate(indptr, indices, uniq, k=k, alpha=0.85, epsilon=1e-4, num_threads=2)
            hs.set_ppr(uniq, tab.len, tab.neigh, tab.score)
            kw.update(k=k, threshold=thr); extra = dict(ppr=tab)
        got = hs.sample(SamplerConfig(aug=aug, **kw), roots=roots, serial_base=3)
        ref = so.sample_batch(indptr, indices, roots, aug=aug, seed=seed, serial_base=3, num_threads=2, **kw, **extra)
        _cmp_batch(ref, got, aug, (trial, kind, n, kw, aug))
        hs.close()
    except Exception as ex:
        bad += 1
        print("MISMATCH/ERROR trial", trial, kind, n, P, num_roots, kw, aug, type(ex).__name__, str(ex)[:300]); sys.stdout.flush()
        if bad >= 5: break
print("done", T, "trials,", bad, "bad")
