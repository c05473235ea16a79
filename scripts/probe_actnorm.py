import torch, time
from shadow_gnn_amd import ops
dev = "cuda:0"
n, F = 289252, 256
torch.manual_seed(0)
def run(nb, act, need_grad=True, iters=20):
    Zs = [torch.randn(n, F, device=dev, requires_grad=need_grad) for _ in range(nb)]
    sc = torch.ones(nb, F, device=dev, requires_grad=True); of = torch.zeros(nb, F, device=dev, requires_grad=True)
    out = ops.act_norm(Zs, [act] * nb, sc, of)
    g = torch.randn_like(out)
    for _ in range(3):
        out.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out.backward(g, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    by = ((2 * nb + 1) if need_grad else (nb + 1)) * 4 * n * F
    print(f"bwd nb={nb} act={act} dZ={need_grad}: {ms:.3f} ms  {by/1e9/(ms/1e3):.0f} GB/s")
    e0.record()
    for _ in range(iters):
        out = ops.act_norm(Zs, [act] * nb, sc, of)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"fwd nb={nb} act={act}: {ms:.3f} ms  {(nb+1)*4*n*F/1e9/(ms/1e3):.0f} GB/s")
for nb in (1, 2):
    for act in ("relu", "elu", "I"):
        run(nb, act)
run(2, "relu", need_grad=False)
