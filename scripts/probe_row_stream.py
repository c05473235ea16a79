"""Memory-system ceiling for the sampler's access pattern: stream the full-graph rows of one benchmark batch's nodes
(products shape, k-hop depth 2 budget 20, 1024 roots) with a bare kernel -- no filter, no LDS -- at several depths /
grid sizes, next to a plain contiguous read of the same number of bytes."""
import sys, numpy as np, torch
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
dev = torch.device("cuda:0")
N, nnz, F, C = SHAPES["products"]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE["products"])
hs = HipSampler(indptr, indices, device=dev, seed=3)
roots = torch.randperm(N, generator=torch.Generator().manual_seed(2)).numpy().astype(np.uint32)
hs.shuffle_targets(roots)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = hs.sample(SamplerConfig(method="khop", depth=2, budget=20), B)
nodes = b.node.contiguous()
ip = indptr.long() & 0xFFFFFFFF
deg = (ip[nodes.long() + 1] - ip[nodes.long()])
nbytes = float(deg.sum()) * 4
print(f"{nodes.numel()} rows, {nbytes / 1e6:.1f} MB of neighbour ids, mean row {float(deg.float().mean()) * 4:.0f} B, median {float(deg.float().median()) * 4:.0f} B")
out = torch.empty(nodes.numel(), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for depth in (1, 2, 4):
    for blocks in (1024, 2048, 4096, 8192):
        for _ in range(3):
            hs._lib.sg_debug_stream_rows(hs._h, nodes.data_ptr(), nodes.numel(), out.data_ptr(), depth, blocks, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hs._lib.sg_debug_stream_rows(hs._h, nodes.data_ptr(), nodes.numel(), out.data_ptr(), depth, blocks, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"depth {depth} blocks {blocks:5d}: {ms * 1e3:7.1f} us  {nbytes / ms / 1e9:7.1f} GB/s")
x = indices[: int(nbytes // 4)]
for _ in range(3): x.sum()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): x.sum()
e1.record(); torch.cuda.synchronize()
print(f"contiguous torch sum of the same bytes: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us  {nbytes / (e0.elapsed_time(e1) / 10) / 1e9:.1f} GB/s")
