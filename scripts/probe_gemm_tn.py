"""A/B of the two weight-gradient kernels (per-wavefront split vs cooperative split), HIP-event timed, interleaved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shadow_gnn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
M = 289309
dZ = torch.randn(M, 768, device=dev, generator=g)[:, :256]
for K in (256, 128):
    X = torch.randn(M, K, device=dev, generator=g)
    for rnd in range(3):
        out = []
        for coop, deep in (("0", "0"), ("1", "0"), ("0", "1")):
            os.environ["SHADOW_GEMM_TN_COOP"] = coop; os.environ["SHADOW_GEMM_TN_DEEP"] = deep
            ops.weight_grad(dZ, X); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.weight_grad(dZ, X)
            e1.record(); torch.cuda.synchronize()
            out.append(f"coop={coop} deep={deep}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
        print(f"K={K} round {rnd}: " + "  ".join(out), flush=True)
