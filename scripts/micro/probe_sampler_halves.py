"""Lead for the sampler's fixed latency: one 1 024-root call against two 512-root calls issued on two streams at once (two
sampler handles over the same graph; nothing merges the halves here): wall time from the first launch to the last kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
dev = torch.device("cuda:0")
N, nnz, F, C = SHAPES["products"]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE["products"])
roots = torch.randperm(N, generator=torch.Generator().manual_seed(2)).numpy().astype(np.uint32)
cfg = SamplerConfig(method="khop", depth=2, budget=20)
hs = [HipSampler(indptr, indices, device=dev, seed=3 + i) for i in range(3)]
for h in hs: h.shuffle_targets(roots)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def one():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hs[0].sample_async(cfg, 1024); e1.record()
    b = hs[0].finish(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), b.num_nodes
def two():
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    cur = torch.cuda.current_stream(dev)
    e0.record()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): hs[1].sample_async(cfg, 512); ea.record()
    with torch.cuda.stream(s2): hs[2].sample_async(cfg, 512); eb.record()
    b1, b2 = hs[1].finish(), hs[2].finish(); torch.cuda.synchronize()
    return max(e0.elapsed_time(ea), e0.elapsed_time(eb)), b1.num_nodes + b2.num_nodes
for name, fn in (("one 1024-root call", one), ("two 512-root calls, two streams", two)):
    for _ in range(3): fn()
    ts = [fn() for _ in range(10)]
    print(f"{name:34s} {np.mean([t for t, _ in ts]) * 1e3:7.1f} us (incl. relocation)   nodes {int(np.mean([n for _, n in ts]))}")
