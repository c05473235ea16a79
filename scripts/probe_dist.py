import numpy as np, torch
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, make_graph_torch
dev = torch.device("cuda:0")
UNCAP, SHAPE = 0, "products"
N, nnz, F, Cc = SHAPES["products"]
from shadow_gnn_amd.synthetic import MAX_DEGREE
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=(None if UNCAP else MAX_DEGREE[SHAPE]))
ip = indptr.cpu().numpy().view(np.uint32).astype(np.int64)
deg = np.diff(ip)
print("deg: mean %.1f max %d p99 %d p999 %d; E[d^2]/E[d]=%.0f" % (deg.mean(), deg.max(), np.percentile(deg,99), np.percentile(deg,99.9), (deg.astype(np.float64)**2).sum()/deg.sum()))
srt = np.sort(deg)[::-1]
print("top10 deg", srt[:10], "sum top100 %.3g top1000 %.3g top10000 %.3g of %.3g" % (srt[:100].sum(), srt[:1000].sum(), srt[:10000].sum(), deg.sum()))
hs = HipSampler(indptr, indices, device=dev, seed=3)
roots = torch.randperm(N, generator=torch.Generator().manual_seed(2))[:1024].numpy().astype(np.uint32)
cfg = SamplerConfig(method="khop", depth=2, budget=20)
b = hs.sample(cfg, roots=roots)
h = b.to_host()
no = h["subg_node_off"].astype(np.int64)
D = np.array([deg[h["node"][no[i]:no[i+1]]].sum() for i in range(1024)])
print("D_s: mean %.3g max %.3g min %.3g p50 %.3g p90 %.3g" % (D.mean(), D.max(), D.min(), np.percentile(D,50), np.percentile(D,90)))
nd = deg[h["node"]]
for thr in (1000, 10000, 50000, 100000):
    print("rows with deg>%d: %d of %d, slot share %.3f" % (thr, (nd>thr).sum(), nd.size, nd[nd>thr].sum()/nd.sum()))
