"""Implicit host <-> device synchronisations in the training steps: torch's sync debug mode over a short bench.py run, unique
warning sites.   usage: python scripts/find_syncs.py [bench.py args]"""
import collections, os, runpy, sys, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
seen = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        seen[f"{filename.split('/')[-1]}:{lineno}"] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
from shadow_gnn_amd import models
orig = models.DeepGNN.step
state = {"n": 0}
def step(self, *a, **k):
    state["n"] += 1
    if state["n"] == 6:
        torch.cuda.set_sync_debug_mode("warn")      # (steady state only: set-up syncs are not the subject)
    return orig(self, *a, **k)
models.DeepGNN.step = step
sys.argv = ["bench.py"] + (sys.argv[1:] or ["--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-tail"])
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
finally:
    torch.cuda.set_sync_debug_mode("default")
    for k, v in seen.most_common(40):
        print(f"{v:5d}  {k}", file=sys.stderr)
