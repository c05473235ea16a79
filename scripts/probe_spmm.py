"""SpMM timing on one sampled products-shape batch, several feature widths (development aid)."""
import sys, numpy as np, torch
from shadow_gnn_amd import ops
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
dev = torch.device("cuda:0")
N, nnz, F0, C = SHAPES["products"]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE["products"])
hs = HipSampler(indptr, indices, device=dev, seed=3)
hs.shuffle_targets(torch.randperm(N, generator=torch.Generator().manual_seed(2))[:4096].numpy().astype(np.uint32))
b = hs.sample(SamplerConfig(method="khop", depth=2, budget=20), 1024)
import os
if os.environ.get("BLOCKDIAG", "1") == "1":
    csr = ops.DeviceCSR(b.indptr, b.indices, subg_off=b.subg_node_off, subg_edge_off=b.subg_edge_off,
                        max_subg_nodes=b.counts["max_subg_nodes"])
else:
    csr = ops.DeviceCSR(b.indptr, b.indices)
adj = ops.adj_norm_rw(csr)
n, e = b.num_nodes, b.num_edges
print("n", n, "e", e)
for F in [int(a) for a in sys.argv[1:]] or [64, 100, 128, 256]:
    X = torch.randn(n, F, device=dev)
    if os.environ.get("PAD", "0") == "1":          # rows on whole 128-byte lines (what LazyRows.gather_dropped leaves)
        Xp = torch.zeros(n, (F + 31) // 32 * 32, device=dev); Xp[:, :F] = X; X = Xp[:, :F]
    for _ in range(3):
        Y = ops.spmm(adj, X)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
        Y = ops.spmm(adj, X)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 50
    alg = 8 * n * F + 4 * (n + 1) + 4 * e
    print(f"F={F:4d}  {ms*1e3:7.1f} us   alg {alg/ms/1e6:7.1f} GB/s   gather-level {(4*e*F + 4*n*F)/ms/1e6:7.1f} GB/s")
