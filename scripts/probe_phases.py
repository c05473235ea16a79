"""Per-phase cycle stamps of the sampler kernel (library built with -DSHADOW_SG_TIMING)."""
import sys, numpy as np, torch, ctypes as C
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
shape = sys.argv[1] if len(sys.argv) > 1 else "products"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
selfe = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
N, nnz, F, Cc = SHAPES[shape]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE[shape])
hs = HipSampler(indptr, indices, device=dev, seed=3)
roots = torch.randperm(N, generator=torch.Generator().manual_seed(2))[:B * 4].numpy().astype(np.uint32)
hs.shuffle_targets(roots)
cfg = SamplerConfig(method="khop", depth=2, budget=20, add_self_edge=bool(selfe))
for _ in range(3):
    b = hs.sample(cfg, B)
buf = np.zeros((B, 16), dtype=np.uint32)
hs._lib.sg_debug_subgraph_stats(hs._h, buf.ctypes.data, B)
st = buf[:, 8:12].astype(np.float64)
d = np.diff(np.concatenate([np.zeros((B, 1)), st], axis=1), axis=1)
names = ["select", "sort+rank+prefix", "scan", "matchsort+write"]
print("per-subgraph cycles (mean / p50 / max):  n=%.0f slots=%.0f e=%.0f" % (buf[:,0].mean(), buf[:,3].mean(), buf[:,1].mean()))
for i, nm in enumerate(names):
    print("  %-16s %9.0f %9.0f %9.0f" % (nm, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
print("  %-16s %9.0f %9.0f %9.0f" % ("total", st[:, 3].mean(), np.median(st[:, 3]), st[:, 3].max()))
ft = buf[:, 12:16].astype(np.float64)
print("  wave0 bulk-path buckets (mean cycles): other/build %.0f  load-wait %.0f  lookup %.0f  emit %.0f" % tuple(ft.mean(0)))
