"""Is the split-bf16 nt GEMM held back by the latency / bandwidth of its A stream?  SHADOW_GEMM_PROBE_A_WRAP=128 makes every
workgroup read the same 128 rows of A (L2-resident) with the identical instruction stream: the difference to the normal
run is what the HBM stream of A costs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shadow_gnn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for M, K in ((289309, 256), (289309, 512), (65536, 256)):
    A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(256, K, device=dev, generator=g) / 16
    def f(): return ops.mm_nt(A, W)
    f(); torch.cuda.synchronize()
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f"M={M} K={K} wrap={os.environ.get('SHADOW_GEMM_PROBE_A_WRAP','0')}: {e0.elapsed_time(e1)/20*1e3:.1f} us", flush=True)
