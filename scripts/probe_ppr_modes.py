"""sg_ppr_push: ordered (bit-exact) vs fifo mode, products shape, k = 200, eps = 1e-5."""
import time, numpy as np, torch
from shadow_gnn_amd.sampler import HipSampler
from shadow_gnn_amd.ppr import ppr_approximate_device
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
dev = torch.device("cuda:0")
N, nnz, F, C = SHAPES["products"]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE["products"])
hs = HipSampler(indptr, indices, device=dev, seed=3)
targets = np.random.default_rng(0).permutation(N)[:16384].astype(np.uint32)
res = {}
for order in ("ordered", "fifo", "ordered", "fifo"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res[order] = ppr_approximate_device(hs, targets, 200, 0.85, 1e-5, order=order)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{order:8s}: {targets.size / dt:9.0f} targets/s ({dt:.2f} s for {targets.size})")
a, b = res["ordered"], res["fifo"]
jac = [len(set(a[1][i, :a[0][i]]) & set(b[1][i, :b[0][i]])) / len(set(a[1][i, :a[0][i]]) | set(b[1][i, :b[0][i]])) for i in range(0, 16384, 64)]
print("top-200 jaccard mean %.3f; root score max |diff| %.2e" % (np.mean(jac), np.abs(a[2][:, 0] - b[2][:, 0]).max()))
