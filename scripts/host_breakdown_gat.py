import atexit, os, runpy, sys, time, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import torch
from shadow_gnn_amd import ops, ops_gat, models, minibatch, tail, optim, dist, layers
acc = collections.defaultdict(lambda: [0, 0.0])
_argv = sys.argv[1:]
_opt = lambda name, dflt: int(_argv[_argv.index(name) + 1]) if name in _argv else dflt
WARM, STEPS = _opt("--warmup", 5), _opt("--steps", 30)
state = {"step": 0}
live = lambda: WARM <= state["step"] < WARM + STEPS
def wrap(owner, name, label=None, static=False):
    fn = getattr(owner, name); label = label or f"{getattr(owner, '__name__', owner)}.{name}"
    def inner(*a, **k):
        on = live(); t0 = time.perf_counter()
        try: return fn(*a, **k)
        finally:
            if on:
                e = acc[label]; e[0] += 1; e[1] += time.perf_counter() - t0
            if label == "DeepGNN.step": state["step"] += 1
    setattr(owner, name, staticmethod(inner) if static else inner)
wrap(models.DeepGNN, "step", "DeepGNN.step"); wrap(models.DeepGNN, "_embed", "DeepGNN._embed"); wrap(models.DeepGNN, "_finish_update", "DeepGNN._finish_update")
wrap(minibatch.MinibatchShallowExtractor, "one_batch", "extractor.one_batch")
wrap(minibatch.MinibatchShallowExtractor, "_top_plan", "extractor._top_plan")
wrap(tail, "build_backward_levels", "tail.build_backward_levels")
wrap(torch.Tensor, "backward", "Tensor.backward")
wrap(ops_gat._GatTail, "forward", static=True); wrap(ops_gat._GatTail, "backward", static=True); wrap(ops_gat._GatTail, "_rows_backward", static=True)
wrap(ops._LinearPair, "forward", static=True); wrap(ops._LinearPair, "backward", static=True); wrap(ops._LinearPair, "_rows_backward", static=True)
wrap(ops, "_an_bwd"); wrap(ops, "_an_fwd"); wrap(ops, "weight_grad"); wrap(ops, "weight_grad_f16_pair")
wrap(layers.GAT, "forward", "GAT.forward")
from shadow_gnn_amd import _lib
_L = _lib.load()
for _nm in dir(_L):
    if _nm.startswith(("sl_", "sg_")) and callable(getattr(_L, _nm)): wrap(_L, _nm, "  C-ABI " + _nm)
@atexit.register
def report():
    steps = max(1, acc["DeepGNN.step"][0])
    for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{k:40s} calls/step {c / steps:5.2f}   ms/step {t / steps * 1e3:7.3f}", file=sys.stderr)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
