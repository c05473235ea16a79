"""Host-side wall time per step of the headline step's pieces (perf_counter wrappers around the Python entry points; the
autograd engine runs backward functions on its own thread, which cProfile does not see).
   usage: python scripts/host_breakdown.py [bench.py arguments]"""
import atexit, os, runpy, sys, time, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from shadow_gnn_amd import ops, models, minibatch, tail, optim, dist   # noqa: F401

if os.environ.get("HB_ROOT_GEMM"):            # (A/B: the row-sparse pass's small products on the library's kernels from this many rows on)
    ops.ROOT_GEMM_MIN_ROWS = int(os.environ["HB_ROOT_GEMM"])
acc = collections.defaultdict(lambda: [0, 0.0])
# only the TIMED region counts (bench.py's warm-up steps come before it, its instrumented steps -- a HIP-event pair and a
# synchronisation around every kernel -- after it): the window is [warmup, warmup + steps) in calls of DeepGNN.step
_argv = sys.argv[1:]
_opt = lambda name, dflt: int(_argv[_argv.index(name) + 1]) if name in _argv else dflt
WARM, STEPS = _opt("--warmup", 5), _opt("--steps", 30)
state = {"step": 0}


def live():
    return WARM <= state["step"] < WARM + STEPS


def wrap(owner, name, label=None, static=False):
    fn = getattr(owner, name)
    label = label or f"{getattr(owner, '__name__', owner)}.{name}"

    def inner(*a, **k):
        on = live()
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            if on:
                e = acc[label]; e[0] += 1; e[1] += time.perf_counter() - t0
            if label == "DeepGNN.step":
                state["step"] += 1
    setattr(owner, name, staticmethod(inner) if static else inner)


# every C-ABI entry (ctypes function pointers raise no profile events)
from shadow_gnn_amd import _lib
_L = _lib.load()
for _nm in dir(_L):
    if _nm.startswith(("sl_", "sg_")) and callable(getattr(_L, _nm)):
        wrap(_L, _nm, "  C-ABI " + _nm)
wrap(ops._SageStack, "backward", static=True)
# HB_LINES="ops._SageStack._sparse_top,models.DeepGNN._finish_update,...": wall time per source line of those functions
_codes, _last = {}, {}


def _line_tracer(frame, event, arg):
    nm = _codes.get(frame.f_code)
    if nm is None:
        return None
    if not live():
        return _line_tracer
    now = time.perf_counter()
    key = id(frame)
    prev = _last.get(key)
    if prev is not None:
        e = acc[f"  {nm}:{prev[0]}"]; e[0] += 1; e[1] += now - prev[1]
    if event == "return":
        _last.pop(key, None)
    else:
        _last[key] = (frame.f_lineno, now)
    return _line_tracer


def trace_lines(owner, name, static=False):
    fn = getattr(owner, name)
    _codes[fn.__code__] = f"{getattr(owner, '__name__', owner)}.{name}"

    def inner(*a, **k):
        old = sys.gettrace()
        sys.settrace(_line_tracer)
        try:
            return fn(*a, **k)
        finally:
            sys.settrace(old)
    setattr(owner, name, staticmethod(inner) if static else inner)


_mods = dict(ops=ops, models=models, minibatch=minibatch, tail=tail, optim=optim, dist=dist)
for spec in filter(None, os.environ.get("HB_LINES", "").split(",")):
    parts = spec.split(".")
    owner = _mods[parts[0]]
    for p_ in parts[1:-1]:
        owner = getattr(owner, p_)
    raw = owner.__dict__.get(parts[-1]) if isinstance(owner, type) else None
    trace_lines(owner, parts[-1], static=isinstance(raw, staticmethod))
wrap(ops._SageStack, "_sparse_top", static=True)
wrap(ops._SageStack, "forward", static=True)
wrap(ops, "_an_bwd")
wrap(ops, "weight_grad")
wrap(ops, "mm_nt")
wrap(ops, "_at_dzn_on_rows")
wrap(tail.TopBackwardPlan, "__init__", "TopBackwardPlan.__init__")
wrap(models.DeepGNN, "step", "DeepGNN.step")
wrap(models.DeepGNN, "_embed", "DeepGNN._embed")
wrap(models.DeepGNN, "_head", "DeepGNN._head")
wrap(minibatch.MinibatchShallowExtractor, "one_batch", "extractor.one_batch")
wrap(models.DeepGNN, "_begin_update", "DeepGNN._begin_update")
wrap(models.DeepGNN, "_finish_update", "DeepGNN._finish_update")
wrap(models.DeepGNN, "_fused_head", "DeepGNN._fused_head")
wrap(models.DeepGNN, "_run_stack", "DeepGNN._run_stack")
wrap(torch.Tensor, "backward", "Tensor.backward")
for cls_name in ("_NodeHead", "_Head", "NodeHead"):
    if hasattr(ops, cls_name):
        wrap(getattr(ops, cls_name), "backward", f"{cls_name}.backward", static=True)
        wrap(getattr(ops, cls_name), "forward", f"{cls_name}.forward", static=True)


@atexit.register
def report():
    steps = max(1, acc["DeepGNN.step"][0])
    print(f"--- host wall time per step ({steps} steps)", file=sys.stderr)
    for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:34s} calls/step {c / steps:5.2f}   ms/step {t / steps * 1e3:7.3f}", file=sys.stderr)


sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
