"""Kernel times of the two GEMM-epilogue launches at the benchmark's shape (the workload of scripts/pmc_fused.sh / ab_fused_depth.sh)."""
import torch
from shadow_gnn_amd import ops
DEV = "cuda"
M = 289000
X = torch.randn(M, 256, device=DEV); AX = torch.randn(M, 256, device=DEV)
Ws = torch.randn(256, 256, device=DEV) * 0.06; Wn = torch.randn(256, 256, device=DEV) * 0.06
sc = torch.ones(2, 256, device=DEV); of = torch.zeros(2, 256, device=DEV)
b = [torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)]
dA = torch.randn(M, 512, device=DEV)
Wcat = torch.randn(256, 512, device=DEV) * 0.06
Zs = [torch.randn(M, 256, device=DEV), torch.randn(M, 256, device=DEV)]
for _ in range(2):
    with ops.KernelTimer() as kt:
        for _ in range(6):
            ops.gemm_act_norm_fwd([X, AX], [Ws, Wn], b, [1, 1], sc, of, 1.0, (0.4, 123))
            ops.gemm_an_bwd(dA, Wcat, Zs, b, [1, 1], sc, of, (0.4, 123))
    torch.cuda.synchronize()
    s = kt.summary()
print(" ".join(f"{k}={v['avg_ms']*1e3:.0f}us" for k, v in sorted(s.items()) if k.startswith("gemm_a")))
