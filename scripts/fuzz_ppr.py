"""Fuzz of sg_ppr_push (ordered mode) against the oracle on extreme graphs: isolated targets, 1-node graphs,
stars, complete graphs, directed paths; tables must be bit-identical.  python scripts/fuzz_ppr.py [seed] [trials]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(seed, trials, verbose=True):
    from oracle import sampler_oracle as so
    from shadow_gnn_amd.ppr import ppr_approximate_device
    from shadow_gnn_amd.sampler import HipSampler
    rng = np.random.default_rng(seed)
    def csr(n, a, b):
        if len(a):
            key = np.unique(np.asarray(a, dtype=np.int64) * n + np.asarray(b, dtype=np.int64))
            rows, cols = key // n, (key % n).astype(np.uint32)
        else:
            rows, cols = np.zeros(0, np.int64), np.zeros(0, np.uint32)
        ip = np.zeros(n + 1, dtype=np.int64); np.add.at(ip, rows + 1, 1)
        return np.cumsum(ip).astype(np.uint32), cols
    failures = []
    for trial in range(trials):
        kind = str(rng.choice(["empty", "complete", "star", "path", "dirpath", "random", "random", "loops"]))
        n = int(rng.choice([1, 2, 3, 9, 40, 65, 300, 2000]))
        if kind in ("star", "path", "dirpath") and n < 2:
            n = 2
        if kind == "empty": ip, ix = csr(n, [], [])
        elif kind == "complete": a, b = np.meshgrid(np.arange(min(n, 80)), np.arange(min(n, 80))); ip, ix = csr(n, a.ravel(), b.ravel())
        elif kind == "star": a = np.zeros(n - 1, int); b = np.arange(1, n); ip, ix = csr(n, np.concatenate([a, b]), np.concatenate([b, a]))
        elif kind == "path": a = np.arange(n - 1); ip, ix = csr(n, np.concatenate([a, a + 1]), np.concatenate([a + 1, a]))
        elif kind == "dirpath": a = np.arange(n - 1); ip, ix = csr(n, a, a + 1)
        elif kind == "loops": a = np.arange(n); ip, ix = csr(n, a, a)
        else:
            m = int(n * rng.choice([0.5, 3, 10])); a = rng.integers(0, n, m); b = rng.integers(0, n, m)
            ip, ix = csr(n, np.concatenate([a, b]), np.concatenate([b, a]))
        T = int(min(n, rng.choice([1, 5, 60])))
        targets = rng.choice(n, T, replace=False).astype(np.uint32)
        k = int(rng.choice([1, 2, 10, 100])); eps = float(rng.choice([1e-2, 1e-4, 1e-6])); alpha = float(rng.choice([0.85, 0.5, 0.99]))
        ctx = (trial, kind, n, T, k, eps, alpha)
        try:
            ref = so.ppr_approximate(ip, ix, targets, k=k, alpha=alpha, epsilon=eps, num_threads=2)
            hs = HipSampler(ip, ix, device=torch.device("cuda:0"), seed=0)
            gl, gn, gs = ppr_approximate_device(hs, targets, k, alpha, eps, hash_slots=1 << 8, num_waves=8)
            hs.close()
            assert np.array_equal(gl, ref.len), ("len", gl[:8], ref.len[:8])
            for i in range(T):
                L = int(gl[i])
                assert np.array_equal(gn[i, :L], ref.neigh[i, :L]), ("neigh", i, gn[i, :L][:8], ref.neigh[i, :L][:8])
                assert np.array_equal(gs[i, :L].view(np.uint32), ref.score[i, :L].view(np.uint32)), ("score", i, gs[i, :L][:4], ref.score[i, :L][:4])
        except Exception as ex:
            failures.append((ctx, type(ex).__name__, str(ex)[:300].replace("\n", " ")))
            if verbose:
                print("BAD", ctx, type(ex).__name__, str(ex)[:300].replace("\n", " ")); sys.stdout.flush()
    return failures


if __name__ == "__main__":
    f = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
    print("done,", len(f), "bad")
