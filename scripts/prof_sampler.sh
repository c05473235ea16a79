#!/usr/bin/env bash
# GPU box: per-kernel times of the sampler pipeline alone (rocprofv3 kernel trace of scripts/probe_sampler_batch.py)
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/prof_sampler"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH="$R"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$R/scripts/probe_sampler_batch.py" "$@" > "$OUT/probe.log" 2>&1
cat "$OUT/probe.log" | tail -12
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:12]:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
