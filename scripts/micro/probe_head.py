"""Stand-alone timing of the fused head (ops.node_head forward + backward) on an idle GPU: r roots, F = 256, C classes.
    SHADOW_HEAD_VARIANT=1|2 python scripts/micro/probe_head.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from shadow_gnn_amd import ops
dev = "cuda:0"
for r, C in ((1024, 47), (128, 47), (256, 40), (1024, 172)):
    F = 256
    lin = torch.nn.Linear(F, C).to(dev)
    sc, of = torch.ones(1, C, device=dev, requires_grad=True), torch.zeros(1, C, device=dev, requires_grad=True)
    lab = torch.randint(0, C, (r,), device=dev)
    emb = torch.randn(r, F, device=dev, requires_grad=True)
    def it():
        loss, *_ = ops.node_head(emb, lin, sc, of, lab)
        loss.backward()
    for _ in range(20): it()
    with ops.KernelTimer() as kt:
        for _ in range(50): it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): it()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 300 * 1e6
    ks = kt.summary()
    print(f"variant {os.environ.get('SHADOW_HEAD_VARIANT', '2')} r={r} C={C}: wall {wall:.1f} us/iter;", {k: round(v['avg_ms'] * 1e3, 1) for k, v in ks.items() if k.startswith('head')})
