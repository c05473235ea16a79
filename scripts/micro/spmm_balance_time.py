"""What the uneven subgraph sizes cost the block-diagonal SpMM: the same number of rows and edges as the benchmark batch in
1 024 blocks of (a) equal size, (b) sizes spread like the k-hop batch (mean 283, std 31), (c) the same sizes sorted
descending (static round-robin then deals every workgroup one block of each size class)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shadow_gnn_amd import ops
dev = torch.device("cuda:0")
def batch(sizes, deg=2.05, seed=0):
    rng = np.random.default_rng(seed)
    noff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(noff[-1])
    rows, cols = [], []
    for a, b in zip(noff[:-1], noff[1:]):
        s = b - a
        m = int(round(deg * s))
        rows.append(a + rng.integers(0, s, m)); cols.append(a + rng.integers(0, s, m))
    r = np.concatenate(rows); c = np.concatenate(cols)
    key = np.unique(r * n + c); r, c = key // n, key % n
    indptr = np.zeros(n + 1, np.int64); np.add.at(indptr, r + 1, 1); indptr = np.cumsum(indptr)
    eoff = indptr[noff]
    t = lambda x: torch.from_numpy(x.astype(np.int32)).to(dev)
    return ops.DeviceCSR(t(indptr), t(c), subg_off=t(noff), subg_edge_off=t(eoff), max_subg_nodes=int(max(sizes)))
rng = np.random.default_rng(1)
spread = np.clip(np.round(rng.normal(283, 31.4, 1024)), 150, 380).astype(np.int64)
for name, sizes in (("equal", np.full(1024, int(spread.mean()))), ("spread", spread), ("spread, sorted descending", np.sort(spread)[::-1].copy())):
    csr = batch(sizes)
    adj = ops.adj_norm_rw(csr)
    X = torch.randn(csr.n, 256, device=dev)
    for _ in range(3): ops.spmm(adj, X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.spmm(adj, X)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} n={csr.n} e={csr.e}  {e0.elapsed_time(e1)/50*1e3:.1f} us")
