#!/usr/bin/env bash
# Every kernel of a benchmark step -- torch's and rocBLAS's included -- with launches per step, average time and ms per
# step (rocprofv3 --kernel-trace --stats over bench.py).  This listing is how the per-parameter gradient adds, the serial
# partial reductions and the one-hot / argmax round trip of the loss were found.
#   usage (GPU box): scripts/kernel_inventory.sh [workload] [steps]
set -uo pipefail
W="${1:-products-khop-sage5}"; K="${2:-30}"
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/inventory_$W"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python "$R/bench.py" --workload "$W" --steps "$K" --warmup 5 --no-cpu-baseline --no-tail > /dev/null 2>&1
python - "$OUT" "$K" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) + 5 + min(int(sys.argv[2]), 10)        # warm-up + timed + instrumented steps
tot = cnt = 0.0
for r in csv.DictReader(open(f)):
    calls = int(r["Calls"])
    if calls < steps * 0.9:
        continue                                                  # set-up kernels (graph generation, ...)
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms; cnt += calls / steps
    if ms > 0.004:
        print(f"{r['Name'][:96]:96s} per-step={calls / steps:5.1f} avg_us={float(r['AverageNs']) / 1e3:7.1f} ms/step={ms:6.3f}")
print(f"sum {tot:.3f} ms/step over {cnt:.0f} launches/step")
PY
