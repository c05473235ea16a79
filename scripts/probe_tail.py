import numpy as np, torch, sys
from shadow_gnn_amd import ops, tail
from shadow_gnn_amd.synthetic import make_graph_torch, SHAPES, MAX_DEGREE
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
dev = torch.device("cuda:0")
N, nnz, F0, C = SHAPES["products"]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE["products"])
hs = HipSampler(indptr, indices, device=dev)
roots = np.random.default_rng(0).permutation(N)[:1024 * 4].astype(np.int64)
cfg = SamplerConfig(method="khop", depth=2, budget=20, add_self_edge=True) if True else None
b = hs.sample(cfg, 1024, roots=roots[:1024])
print("batch", b.num_nodes, b.num_edges); sys.stdout.flush()
csr = ops.DeviceCSR(b.indptr, b.indices, subg_off=b.subg_node_off, subg_edge_off=b.subg_edge_off, max_subg_nodes=b.counts["max_subg_nodes"])
levels = tail.build_tail_plan(csr, b.target, 5, eager_transpose=True)
torch.cuda.synchronize(); print("levels", [(lv.r, lv.m_in, int(lv.indices.numel())) for lv in levels]); sys.stdout.flush()
adj = ops.adj_norm_rw(csr, dropedge=0.05)
F = 256
X = torch.randn(csr.n, F, device=dev, requires_grad=True)
x = X
for i, lv in enumerate(levels):
    assert int(lv.indices.max()) < lv.m_in and int(lv.self_idx.max()) < lv.m_in and int(lv.indptr[-1]) == lv.indices.numel()
    xs, ax = tail.rect_gather_spmm(x, lv, adj)
    torch.cuda.synchronize(); print("fwd level", i, xs.shape, ax.shape, float(ax.abs().sum())); sys.stdout.flush()
    x = xs + ax
(x.sum()).backward()
torch.cuda.synchronize(); print("bwd ok", float(X.grad.abs().sum())); sys.stdout.flush()
