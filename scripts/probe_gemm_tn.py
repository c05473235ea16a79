"""A/B of the weight-gradient kernels, HIP-event timed, interleaved: per-wavefront split, cooperative split with the plain
step loop, cooperative split with the interleaved step (SHADOW_GEMM_TN_PIPE); checks that all three agree bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shadow_gnn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for M in (289309, 39500, 1000):
    dZ = torch.randn(M, 768, device=dev, generator=g)[:, :256]
    for K in (256, 128):
        X = torch.randn(M, K, device=dev, generator=g)
        for rnd in range(2):
            out, res = [], []
            for coop, pipe in (("0", "0"), ("1", "0"), ("1", "1")):
                os.environ["SHADOW_GEMM_TN_COOP"] = coop; os.environ["SHADOW_GEMM_TN_PIPE"] = pipe
                res.append(ops.weight_grad(dZ, X)); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): ops.weight_grad(dZ, X)
                e1.record(); torch.cuda.synchronize()
                out.append(f"coop={coop} pipe={pipe}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
            same = all(torch.equal(res[0], r) for r in res[1:])
            print(f"M={M} K={K} round {rnd}: " + "  ".join(out) + f"  identical={same}", flush=True)
