"""Static round-robin assignment of (subgraph, tile-group) items to the block-diagonal SpMM's workgroups on one benchmark
batch: how uneven the per-workgroup row totals are (the kernel ends with its slowest workgroup)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
dev = torch.device("cuda:0")
N, nnz, F0, C = SHAPES["products"]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE["products"])
hs = HipSampler(indptr, indices, device=dev, seed=3)
hs.shuffle_targets(torch.randperm(N, generator=torch.Generator().manual_seed(2))[:4096].numpy().astype(np.uint32))
b = hs.sample(SamplerConfig(method="khop", depth=2, budget=20), 1024)
off = b.subg_node_off.cpu().numpy().astype(np.int64)
sz = np.diff(off)
print("subgraphs", sz.size, "rows mean %.1f std %.1f min %d max %d" % (sz.mean(), sz.std(), sz.min(), sz.max()))
grid, tg, groups = 512, 4, 2
items = np.repeat(sz, groups)                      # item gi = subgraph gi // groups
per_wg = np.zeros(grid)
for gi, rows in enumerate(items):
    per_wg[gi % grid] += rows * tg                # tg tiles per item
print("static: per-workgroup tile-rows mean %.0f max %.0f  max/mean %.3f" % (per_wg.mean(), per_wg.max(), per_wg.max() / per_wg.mean()))
# greedy dynamic (next free workgroup takes the next item, in order)
t = np.zeros(grid)
for rows in items:
    i = t.argmin(); t[i] += rows * tg
print("dynamic in order: max/mean %.3f" % (t.max() / t.mean()))
t = np.zeros(grid)
for rows in sorted(items, reverse=True):
    i = t.argmin(); t[i] += rows * tg
print("dynamic largest first: max/mean %.3f" % (t.max() / t.mean()))
