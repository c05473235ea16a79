import torch, time
dev="cuda:0"
n, Fi, Fo = 289252, 256, 256
X = torch.randn(n, Fi, device=dev); dZ = torch.randn(n, Fo, device=dev); W = torch.randn(Fo, Fi, device=dev)
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
fl = 2*n*Fi*Fo
ms = t(lambda: dZ.t() @ X); print(f"dW plain mm: {ms:.3f} ms {fl/ms/1e9:.1f} TF")
ms = t(lambda: torch.nn.functional.linear(X, W)); print(f"fwd linear: {ms:.3f} ms {fl/ms/1e9:.1f} TF")
ms = t(lambda: dZ @ W); print(f"dX mm: {ms:.3f} ms {fl/ms/1e9:.1f} TF")
ref = dZ.t() @ X
for S in (32, 64, 128, 256, 512):
    c = n // S
    def f():
        p = torch.bmm(dZ[:S*c].view(S, c, Fo).transpose(1, 2), X[:S*c].view(S, c, Fi))
        return p.sum(0) + dZ[S*c:].t() @ X[S*c:]
    ms = t(f); err = (f()-ref).abs().max().item()/ref.abs().max().item()
    print(f"dW split S={S}: {ms:.3f} ms {fl/ms/1e9:.1f} TF relerr {err:.2e}")
# bf16x3
def split(a):
    h = a.bfloat16(); l = (a - h.float()).bfloat16(); return h, l
def bf3():
    Xh, Xl = split(X); Wh, Wl = split(W)
    return (torch.mm(Xh, Wh.t(), out_dtype=torch.float32) + torch.mm(Xh, Wl.t(), out_dtype=torch.float32) + torch.mm(Xl, Wh.t(), out_dtype=torch.float32))
try:
    ms = t(bf3); r = torch.nn.functional.linear(X, W); err=(bf3()-r).abs().max().item()/r.abs().max().item()
    print(f"fwd bf16x3 (incl. splits): {ms:.3f} ms relerr {err:.2e}")
    Xh, Xl = split(X); Wh, Wl = split(W)
    ms = t(lambda: torch.mm(Xh, Wh.t(), out_dtype=torch.float32)); print(f"one bf16 mm fp32 out: {ms:.3f} ms {fl/ms/1e9:.1f} TF")
except Exception as ex:
    print("bf16 out_dtype path failed:", type(ex).__name__, str(ex)[:200])
