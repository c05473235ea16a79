import sys, os, numpy as np, torch
from shadow_gnn_amd import ops
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
dev = torch.device("cuda:0")
N, nnz, F0, C = SHAPES["products"]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE["products"])
hs = HipSampler(indptr, indices, device=dev, seed=3)
roots = torch.randperm(N, generator=torch.Generator().manual_seed(2)).numpy().astype(np.uint32)
hs.shuffle_targets(roots)
b = hs.sample(SamplerConfig(method="khop", depth=2, budget=20), 1024)
csr = ops.DeviceCSR(b.indptr, b.indices, subg_off=b.subg_node_off, subg_edge_off=b.subg_edge_off, max_subg_nodes=b.counts["max_subg_nodes"])
adj = ops.adj_norm_rw(csr, dropedge=0.05)
n = b.num_nodes
feat = torch.randn(N, F0, device=dev)
def timeit(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
e = csr.e
for F in (100, 128, 256):
    X = torch.randn(n, F, device=dev)
    us = timeit(lambda: ops.spmm(adj, X))
    by = 4*(n+1) + 8*e + 8*n*F
    print(f"spmm F={F}: {us:.1f} us  {by/us/1e6:.2f} TB/s alg  ({by/1e6:.0f} MB)")
lazy = ops.LazyRows(feat, b.node)
for p in (0.0, 0.4):
    for wd in (True, False):
        us = timeit(lambda: ops.spmm_gather(adj, lazy, drop_p=p, want_dense=wd))
        by = 4*(n+1) + 8*e + 4*n + 4*n*F0*(2 + (1 if wd else 0))
        print(f"spmm_gather F=100 p={p} dense_copy={wd}: {us:.1f} us  {by/us/1e6:.2f} TB/s alg ({by/1e6:.0f} MB)")
us = timeit(lambda: ops.gather_rows(feat, b.node)); print(f"gather F=100: {us:.1f} us")
Xg = ops.gather_rows(feat, b.node)
us = timeit(lambda: torch.nn.functional.dropout(Xg, 0.4, True)); print(f"torch dropout F=100: {us:.1f} us")
xd, _ = lazy.gather_dropped(0.4)
us = timeit(lambda: lazy.gather_dropped(0.4)); print(f"gather+dropout into padded rows F=100: {us:.1f} us")
us = timeit(lambda: ops.spmm(adj, xd)); print(f"spmm F=100 on 128-float row pitch: {us:.1f} us  {(4*(n+1)+8*e+8*n*100)/us/1e6:.2f} TB/s alg")
