"""Seeded synthetic graphs of the benchmark shapes:
uint32 CSR, symmetric, deduplicated, sorted rows, no self-loops, heavy-tailed
degrees (one endpoint uniform, the other Pareto(1.5)-weighted)."""
import numpy as np

SHAPES = {
    # name: (num_nodes, target nnz (directed entries), feature width, classes)
    "arxiv": (169_343, 2_331_418, 128, 40),
    "products": (2_449_029, 123_718_280, 100, 47),
    "papers100M": (111_059_956, 3_231_371_744, 128, 172),
}

# Largest node degree of the real (undirected) OGB graphs.  An uncapped
# Pareto(1.5) weight vector over 2.4 M nodes puts >1 M edges on a single node
# (60x the real products maximum) and the benchmark degenerates into scanning
# four rows; the generators clip the weights so that the expected maximum
# degree matches the dataset the shape is named after.
MAX_DEGREE = {"arxiv": 13_161, "products": 17_481, "papers100M": 250_000}


def make_graph_numpy(n, avg_deg, seed=0):
    """Small/medium graphs on the host (tests, smoke)."""
    rng = np.random.default_rng(seed)
    m = int(n * avg_deg // 2)
    w = rng.pareto(1.5, n) + 1.0
    w /= w.sum()
    a = rng.integers(0, n, m)
    b = rng.choice(n, size=m, p=w)
    keep = a != b
    a, b = a[keep], b[keep]
    key = np.concatenate([a * n + b, b * n + a])
    key = np.unique(key)
    rows = key // n
    cols = (key % n).astype(np.uint32)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    indptr = np.cumsum(indptr).astype(np.uint32)
    return indptr, cols


def make_graph_torch(n, nnz_target, seed=0, device="cuda", max_degree=None):
    """Benchmark-scale graphs generated on the GPU (sort-based symmetrise +
    dedupe).  Returns int32 tensors (uint32 bit patterns) indptr[n+1], indices[nnz]."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    m = nnz_target // 2
    # Pareto(1.5)+1 weights via inverse CDF; weighted endpoint via searchsorted on the CDF
    u = torch.rand(n, generator=g, device=device, dtype=torch.float64).clamp_(min=1e-12)
    w = u.pow_(-1.0 / 1.5)
    if max_degree is not None:
        for _ in range(3):      # clip so that E[deg] of the heaviest node ~ max_degree
            w.clamp_(max=float(max_degree) * float(w.sum()) / m)
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    a = torch.randint(0, n, (m,), generator=g, device=device, dtype=torch.int64)
    r = torch.rand(m, generator=g, device=device, dtype=torch.float64)
    b = torch.searchsorted(cdf, r).clamp_(max=n - 1)
    del r, cdf, w, u
    keep = a != b
    a, b = a[keep], b[keep]
    k1, k2 = a * n + b, b * n + a
    del a, b, keep
    if 2 * k1.numel() < (1 << 30):
        key = torch.unique(torch.cat([k1, k2]))      # sorted + deduplicated
        del k1, k2
    else:
        # torch's sort / boolean indexing take < 2^31 elements: de-duplicate row ranges separately (keys are
        # row-major, so the sorted pieces concatenate into the sorted whole)
        parts = -(-2 * k1.numel() // (1 << 29))
        pieces = []
        for i in range(parts):
            lo, hi = (i * n // parts) * n, (((i + 1) * n // parts) * n if i + 1 < parts else (n + 1) * n)
            sel = torch.cat([k1[(k1 >= lo) & (k1 < hi)], k2[(k2 >= lo) & (k2 < hi)]])
            pieces.append(torch.unique(sel))
            del sel
        del k1, k2
        key = torch.cat(pieces)
        del pieces
    rows = torch.div(key, n, rounding_mode="floor")
    cols = (key - rows * n).to(torch.int32)
    del key
    counts = torch.bincount(rows, minlength=n)
    del rows
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    assert int(indptr[-1]) < 2 ** 32
    # uint32 bit pattern in an int32 tensor
    indptr32 = (indptr & 0xFFFFFFFF).to(torch.int64)
    indptr32 = torch.where(indptr32 >= 2 ** 31, indptr32 - 2 ** 32, indptr32).to(torch.int32)
    return indptr32, cols
