"""What the fused dropout hash costs the two GEMM-epilogue launches: the same launches with drop_p = 0.4 and 0 (benchmark shape)."""
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
for drop in ((0.4, 123), (0.0, 0)):
    for _ in range(2):
        with ops.KernelTimer() as kt:
            for _ in range(6):
                ops.gemm_act_norm_fwd([X, AX], [Ws, Wn], b, [1, 1], sc, of, 1.0, drop)
                ops.gemm_an_bwd(dA, Wcat, Zs, b, [1, 1], sc, of, drop)
        torch.cuda.synchronize()
        s = kt.summary()
    print("drop", drop[0], " ".join(f"{k}={v['avg_ms']*1e3:.0f}us" for k, v in sorted(s.items()) if k.startswith("gemm_a")))
