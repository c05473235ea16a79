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
import os
if os.environ.get("METHOD", "khop") == "ppr":      # configs[2]: top-200 PPR subgraphs (~150 rows, ~2 edges per row)
    from shadow_gnn_amd.ppr import ppr_approximate_device
    uniq = np.unique(np.random.default_rng(1).permutation(N)[:1024]).astype(np.uint32)
    ln, nb, sc = ppr_approximate_device(hs, uniq, 200, 0.85, 1e-5)
    hs.set_ppr(uniq, ln, nb, sc)
    hs.shuffle_targets(uniq)
    b = hs.sample(SamplerConfig(method="ppr", k=200, threshold=0.0), 1024)
else:
    b = hs.sample(SamplerConfig(method="khop", depth=2, budget=20), 1024)
cap = int(os.environ.get("DEGCAP", "0"))
if cap:                                           # (what do the long rows cost?  the same batch with every row cut to `cap` entries)
    ip = b.indptr.cpu().numpy().astype(np.int64); ix = b.indices.cpu().numpy()
    deg = np.minimum(np.diff(ip), cap)
    print("rows", len(deg), "max degree", int(np.diff(ip).max()), "rows above cap", int((np.diff(ip) > cap).sum()))
    nip = np.concatenate([[0], np.cumsum(deg)])
    keep = np.concatenate([ix[ip[i]:ip[i] + deg[i]] for i in range(len(deg))]) if len(deg) else ix[:0]
    so = b.subg_node_off.cpu().numpy().astype(np.int64)
    b.indptr = torch.from_numpy(nip.astype(np.int32)).to(dev); b.indices = torch.from_numpy(keep.astype(np.int32)).to(dev)
    b.subg_edge_off = torch.from_numpy(nip[so].astype(np.int32)).to(dev)
    b.counts["e_tot"] = int(nip[-1])
if os.environ.get("BLOCKDIAG", "1") == "1":
    csr = ops.DeviceCSR(b.indptr, b.indices, subg_off=b.subg_node_off, subg_edge_off=b.subg_edge_off,
                        max_subg_nodes=b.counts["max_subg_nodes"])
else:
    csr = ops.DeviceCSR(b.indptr, b.indices)
adj = ops.adj_norm_rw(csr)
n, e = b.num_nodes, int(b.indices.numel())
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
