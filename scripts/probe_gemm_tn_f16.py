"""Weight-gradient kernel on two fp16 pieces (sl_gemm_tn_f16) against the bf16 x 3 kernel: error relative to sum |a||b|
against fp64 on several operand distributions, determinism, timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shadow_gnn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
def err(got, dZ, X):
    ref = dZ.double().t() @ X.double()
    den = dZ.abs().double().t() @ X.abs().double()
    e = (got.double() - ref).abs() / den.clamp_min(1e-300)
    return float(e.max()), float((e * e).mean().sqrt())
def cases(M):
    dZ = torch.randn(M, 768, device=dev, generator=g)[:, :256]; X = torch.randn(M, 256, device=dev, generator=g)
    yield "unit normal", dZ, X
    yield "rows over 30 binades", dZ * torch.exp2(torch.randint(-15, 15, (M, 1), device=dev, generator=g).float()), X * torch.exp2(torch.randint(-15, 15, (M, 1), device=dev, generator=g).float())
    yield "e^N(0,4) inside rows", dZ * torch.exp(2 * torch.randn(M, 256, device=dev, generator=g)), X * torch.exp(2 * torch.randn(M, 256, device=dev, generator=g))
    mask = (torch.rand(M, 1, device=dev, generator=g) < 0.004).float()
    yield "0.4 % non-zero rows of dZ", dZ * mask, X
    Xc = X.clone(); Xc[:, 7] *= 1e4
    yield "dropout zeros + one column 1e4", dZ * (torch.rand(M, 256, device=dev, generator=g) > 0.4).float(), Xc
for M in (289309, 40000, 5000):
    for name, dZ, X in cases(M):
        dZ = dZ.contiguous() if dZ.stride(1) != 1 else dZ
        f16 = ops.weight_grad_f16(dZ, X)
        bf = ops.weight_grad(dZ, X)
        same = torch.equal(f16, ops.weight_grad_f16(dZ, X))
        print(f"M={M:7d} {name:32s} fp16x2 max/rms {err(f16, dZ, X)[0]:.2e} / {err(f16, dZ, X)[1]:.2e}   bf16x3 {err(bf, dZ, X)[0]:.2e} / {err(bf, dZ, X)[1]:.2e}   deterministic {same}  finite {bool(torch.isfinite(f16).all())}", flush=True)
M = 289309
dZ = torch.randn(M, 768, device=dev, generator=g)[:, :256]; X = torch.randn(M, 256, device=dev, generator=g)
da, xa = ops.row_amax(dZ), ops.row_amax(X)
buf = torch.randn(M, 768, device=dev, generator=g)
ja = ops.row_amax(buf[:, :512])
p1, p2 = ops.weight_grad_f16_pair(buf[:, :256], buf[:, 256:512], X, ja, xa)
print("pair == two launches:", torch.equal(p1, ops.weight_grad_f16(buf[:, :256], X, ja, xa)), torch.equal(p2, ops.weight_grad_f16(buf[:, 256:512], X, ja, xa)))
for rnd in range(2):
    out = []
    for name, fn in (("bf16x3", lambda: ops.weight_grad(dZ, X)), ("fp16x2", lambda: ops.weight_grad_f16(dZ, X, da, xa)),
                     ("two fp16x2 launches", lambda: (ops.weight_grad_f16(buf[:, :256], X, ja, xa), ops.weight_grad_f16(buf[:, 256:512], X, ja, xa))),
                     ("fp16x2 pair", lambda: ops.weight_grad_f16_pair(buf[:, :256], buf[:, 256:512], X, ja, xa))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{name}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
    print("M=289309 " + "  ".join(out), flush=True)
