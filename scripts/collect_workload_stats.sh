#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of bench.py for the other BASELINE workloads (kernel time only; the PMC passes of
# scripts/collect_profiles.sh are for the default workload).  Run on the GPU box; copies land in gpurun_out/prof_<tag>_w/.
#   usage: scripts/collect_workload_stats.sh <round-tag>
TAG="${1:-r03}"
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/prof_${TAG}_w"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in products-ppr-sage5 products-khop3-gat5 arxiv-khop-sage5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w" -- python "$R/bench.py" --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-tail > "$OUT/$w.log" 2>&1
  find "$OUT/$w" -name "*kernel_trace.csv" -delete; find "$OUT/$w" -name "*.db" -delete
  cp "$(find "$OUT/$w" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats_$w.csv"
done
ls -la "$OUT"/*.csv
