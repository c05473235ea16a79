"""Accuracy and speed of the split-bf16 MFMA GEMM vs rocBLAS fp32 (development aid)."""
import torch, sys
from shadow_gnn_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, K, N) in [(289252, 256, 256), (289252, 100, 256), (289252, 256, 47), (40000, 128, 256), (1000003, 256, 256)]:
    A = torch.randn(M, K, device=dev) * torch.rand(M, 1, device=dev) * 3
    W = torch.randn(N, K, device=dev) * 0.1
    ref = (A[:20000].double() @ W.double().t())
    got = ops.mm_nt(A, W)
    blas = A @ W.t()
    den = (A[:20000].abs().double() @ W.abs().double().t())
    e_split = ((got[:20000].double() - ref).abs() / den).max().item()
    e_blas = ((blas[:20000].double() - ref).abs() / den).max().item()
    ms_s = t(lambda: ops.mm_nt(A, W)); ms_b = t(lambda: A @ W.t())
    fl = 2.0 * M * K * N
    print(f"M={M} K={K} N={N}: split {ms_s:.3f} ms ({fl/ms_s/1e9:.0f} TF-equiv, {4*M*(K+N)/ms_s/1e6:.0f} GB/s)  rocBLAS {ms_b:.3f} ms ({fl/ms_b/1e9:.0f} TF)  "
          f"max err/|a||b|: split {e_split:.2e} rocBLAS {e_blas:.2e}  maxabs diff {float((got-blas).abs().max()):.3e}")
print("--- weight gradient dW = dZ^T X")
for (M, N, K) in [(289252, 256, 256), (289252, 256, 100)]:
    dZ = torch.randn(M, N, device=dev); X = torch.randn(M, K, device=dev)
    ops.GEMM_SPLIT = True;  ms_s = t(lambda: ops.weight_grad(dZ, X)); a = ops.weight_grad(dZ, X)
    ops.GEMM_SPLIT = False; ms_b = t(lambda: ops.weight_grad(dZ, X)); b = ops.weight_grad(dZ, X)
    ops.GEMM_SPLIT = True
    ref = dZ.double().t() @ X.double(); den = dZ.abs().double().t() @ X.abs().double()
    print(f"M={M} N={N} K={K}: split {ms_s:.3f} ms  rocBLAS split-K {ms_b:.3f} ms  err split {float(((a.double()-ref).abs()/den).max()):.2e} rocBLAS {float(((b.double()-ref).abs()/den).max()):.2e}")
