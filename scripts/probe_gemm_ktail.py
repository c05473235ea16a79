"""K-tail GEMM (K = 100) at the benchmark's height: input rows on a 100-float pitch (plain gather) vs a 128-float
pitch (LazyRows.gather_dropped), and the same from a GAT-like caller (development aid)."""
import torch
from shadow_gnn_amd import ops
dev = torch.device("cuda:0")
M = 295000
W = torch.randn(256, 100, device=dev)
A100 = torch.randn(M, 100, device=dev)
A128 = torch.randn(M, 128, device=dev)[:, :100]
for name, A in (("pitch 100", A100), ("pitch 128", A128)):
    for _ in range(3): ops.mm_nt(A, W)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.mm_nt(A, W)
    e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (incl. pack)")
