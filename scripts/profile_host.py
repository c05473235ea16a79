"""cProfile of the host side of the train step (development aid): where the Python time of a step goes."""
import cProfile, pstats, io, sys, os, runpy
sys.argv = ["bench.py", "--steps", "60", "--warmup", "5", "--no-cpu-baseline", "--no-tail"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
