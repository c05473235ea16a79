"""GEMM-epilogue forward kernel at the layer-0 shape: K = 100 (tail unit, predicated A loads) against the same rows read as
K = 128 with zero pads (no tail): what the tail handling costs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shadow_gnn_amd import ops
DEV = "cuda"
M = 289000
Xp = torch.zeros(M, 128, device=DEV); AXp = torch.zeros(M, 128, device=DEV)
Xp[:, :100] = torch.randn(M, 100, device=DEV); AXp[:, :100] = torch.randn(M, 100, device=DEV)
W100 = [torch.randn(256, 100, device=DEV) * 0.06 for _ in range(2)]
W128 = [torch.zeros(256, 128, device=DEV) for _ in range(2)]
for a, b in zip(W128, W100): a[:, :100] = b
sc = torch.ones(2, 256, device=DEV); of = torch.zeros(2, 256, device=DEV)
b = [torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)]
res = {}
for name, Xs, Ws in (("K=100 tail", [Xp[:, :100], AXp[:, :100]], W100), ("K=128", [Xp, AXp], W128)):
    for _ in range(2):
        with ops.KernelTimer() as kt:
            for _ in range(6):
                Z, out = ops.gemm_act_norm_fwd(Xs, Ws, b, [1, 1], sc, of, 1.0, (0.4, 123))
        torch.cuda.synchronize()
        s = kt.summary()
    res[name] = out
    print(name, " ".join(f"{k}={v['avg_ms']*1e3:.0f}us" for k, v in sorted(s.items()) if k.startswith("gemm_a")))
print("identical outputs:", torch.equal(res["K=100 tail"], res["K=128"]))
