"""fp16 two-piece split of the GEMM-epilogue kernels: error against fp64 next to the six-term bf16 kernel and a plain
fp32 GEMM, and kernel times at the benchmark's shape.  Usage: PYTHONPATH=. python scripts/probe_f16_split.py"""
import torch
from shadow_gnn_amd import ops

DEV = "cuda"


def errs(name, got, A, W):
    ref = A.double() @ W.double().t()
    den = A.abs().double() @ W.abs().double().t() + 1e-300
    e = (got.double() - ref).abs() / den
    print(f"  {name:28s} max {e.max().item():.3e}  rms {e.pow(2).mean().sqrt().item():.3e}")
    return e.max().item()


def fused_z(A, W):
    N = W.shape[0]
    sc = torch.ones(1, N, device=DEV); of = torch.zeros(1, N, device=DEV)
    Zs, _ = ops.gemm_act_norm_fwd([A], [W], [None], [0], sc, of, 1.0, (0.0, 0))
    return Zs[0]


def case(title, A, W):
    print(title, tuple(A.shape), tuple(W.shape))
    errs("fp16 x2, 3 terms (fused)", fused_z(A, W), A, W)
    errs("bf16 x3, 6 terms (mm_nt)", ops.mm_nt(A, W), A, W)
    errs("torch fp32 matmul", A @ W.t(), A, W)


g = torch.Generator(device=DEV).manual_seed(1)
M, K, N = 4096, 256, 256
A = torch.randn(M, K, device=DEV, generator=g)
W = torch.randn(N, K, device=DEV, generator=g) * 0.06
case("unit normal", A, W)
A2 = A * torch.exp(torch.randn(M, 1, device=DEV, generator=g) * 8)          # rows over ~30 binades
W2 = W * torch.exp(torch.randn(N, 1, device=DEV, generator=g) * 4)
case("row magnitudes e^N(0,8)", A2, W2)
A3 = A * torch.exp(torch.randn(M, K, device=DEV, generator=g) * 4)          # 17+ binades INSIDE a row
case("element magnitudes e^N(0,4) inside rows", A3, W)
A4 = A.clone(); A4[torch.rand(M, K, device=DEV, generator=g) < 0.4] = 0; A4[:, 7] = 1e4   # dropout zeros + one hot column
case("dropout zeros + outlier column", A4, W)
A5 = torch.randn(M, 100, device=DEV, generator=g); W5 = torch.randn(N, 100, device=DEV, generator=g)
case("K = 100 (tail)", torch.nn.functional.pad(A5, (0, 28))[:, :100], W5)
A6 = torch.randn(M, 512, device=DEV, generator=g) * 1e-6; W6 = torch.randn(N, 512, device=DEV, generator=g)
case("K = 512, tiny gradients", A6, W6)

# ---- times at the benchmark's shape
M = 289000
X = torch.randn(M, 256, device=DEV); AX = torch.randn(M, 256, device=DEV)
Ws = torch.randn(256, 256, device=DEV) * 0.06; Wn = torch.randn(256, 256, device=DEV) * 0.06
sc = torch.ones(2, 256, device=DEV); of = torch.zeros(2, 256, device=DEV)
b = [torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)]
for _ in range(3):
    with ops.KernelTimer() as kt:
        for _ in range(5):
            Zs, out = ops.gemm_act_norm_fwd([X, AX], [Ws, Wn], b, [1, 1], sc, of, 1.0, (0.4, 123))
            dA = torch.randn(M, 512, device=DEV)
            Wcat = torch.randn(256, 512, device=DEV) * 0.06
            ops.gemm_an_bwd(dA, Wcat, Zs, b, [1, 1], sc, of, (0.4, 123))
            ops.row_amax(X)
            ops.row_amax(dA)
for k, v in sorted(kt.summary().items()):
    print(k, {a: (round(x, 4) if isinstance(x, float) else x) for a, x in v.items()})
