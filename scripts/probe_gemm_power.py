import torch
from shadow_gnn_amd import ops
dev="cuda:0"
def t(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M,K,N=289252,256,256
for name, A, W in [("random", torch.randn(M,K,device=dev), torch.randn(N,K,device=dev)*0.1),
                   ("zeros", torch.zeros(M,K,device=dev), torch.zeros(N,K,device=dev)),
                   ("A rand, W zero", torch.randn(M,K,device=dev), torch.zeros(N,K,device=dev)),
                   ("A zero, W rand", torch.zeros(M,K,device=dev), torch.randn(N,K,device=dev)),
                   ("bf16-exact values", torch.randn(M,K,device=dev).bfloat16().float(), torch.randn(N,K,device=dev).bfloat16().float())]:
    print(f"{name}: split {t(lambda: ops.mm_nt(A,W)):.3f} ms   rocBLAS {t(lambda: A@W.t()):.3f} ms")
