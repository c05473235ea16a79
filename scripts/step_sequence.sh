#!/usr/bin/env bash
# The kernels of ONE steady-state step in launch order (all queues), from a rocprofv3 kernel trace of bench.py.
#   usage: scripts/step_sequence.sh <tag> [bench args...]
set -uo pipefail
TAG="${1:-seq}"; shift || true
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/seq_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python "$R/bench.py" --steps 12 --warmup 5 --no-cpu-baseline --no-tail --no-other-workloads "$@" > "$OUT/bench.log" 2>&1
python - "$OUT" <<'PY' > "$OUT/sequence.txt"
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows)
adam = [i for i, e in enumerate(ev) if "clip_adam" in e[2]]
# the step between the 10th and 11th optimiser launches (a step WITHOUT the four-step sampler call when possible: take the shorter of two)
best = None
for k in (9, 10, 11, 12):
    if k + 1 < len(adam):
        seg = ev[adam[k] + 1: adam[k + 1] + 1]
        if best is None or len(seg) < len(best): best = seg
short = lambda k: (k[k.find("shadow::") + 8:] if "shadow::" in k else k)[:110]
t0 = best[0][0]
for s, e, k, q in best:
    print(f"{(s - t0) / 1e3:9.1f} us  q{q}  {(e - s) / 1e3:8.1f} us  {short(k)}")
PY
rm -rf "$OUT/trace"
tail -3 "$OUT/sequence.txt"
