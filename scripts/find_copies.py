"""Which ops copy whole [n, F] tensors in a training step?  torch.profiler over a short bench.py run: aten::copy_ / clone / contiguous
calls with their input shapes and the Python source line that issued them.   usage: python scripts/find_copies.py [bench.py args]"""
import os, runpy, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity

sys.argv = ["bench.py"] + (sys.argv[1:] or ["--workload", "products-ppr-sage5", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-tail"])
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    try:
        runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
    except SystemExit:
        pass
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60), file=sys.stderr)
rows = []
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add_", "aten::add", "aten::zeros", "aten::zero_", "aten::fill_") and e.input_shapes:
        big = [s for s in e.input_shapes if s and len(s) == 2 and s[0] * s[1] >= 8_000_000]
        if big:
            st = [f for f in (e.stack or []) if "shadow_gnn_amd" in f or "bench.py" in f]
            rows.append((e.name, str(big[0]), st[0] if st else (e.stack[0] if e.stack else "?"), e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total))
from collections import Counter
c = Counter((r[0], r[1], r[2]) for r in rows)
t = Counter()
for r in rows:
    t[(r[0], r[1], r[2])] += r[3]
for k, v in sorted(c.items(), key=lambda kv: -t[kv[0]]):
    print(f"{v:4d} x {k[0]:18s} {k[1]:18s} dev us total {t[k]:10.0f}   {k[2]}", file=sys.stderr)
