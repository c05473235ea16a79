"""Practical HBM rates on this box for calibration: device copy, read-only reduction, write-only fill."""
import torch
dev = "cuda:0"
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (296, 1184, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a); c = torch.randn(n, device=dev)
    ms = t(lambda: b.copy_(a)); print(f"{mb} MiB copy      : {ms:.3f} ms  {2 * n * 4 / ms / 1e6:.0f} GB/s (read + write)")
    ms = t(lambda: torch.add(a, c, out=b)); print(f"{mb} MiB add 2r+1w : {ms:.3f} ms  {3 * n * 4 / ms / 1e6:.0f} GB/s")
    ms = t(lambda: a.sum()); print(f"{mb} MiB sum (read): {ms:.3f} ms  {n * 4 / ms / 1e6:.0f} GB/s")
    ms = t(lambda: b.fill_(1.0)); print(f"{mb} MiB fill      : {ms:.3f} ms  {n * 4 / ms / 1e6:.0f} GB/s")
