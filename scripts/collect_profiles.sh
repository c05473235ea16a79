#!/usr/bin/env bash
# Run on the GPU box (gpurun): rocprofv3 kernel-trace summary + separate PMC passes
# (FETCH_SIZE / WRITE_SIZE, never combined with other trace domains) of bench.py,
# summarised into profiles/ by scripts/summarize_profiles.py.
#   usage: scripts/collect_profiles.sh <round-tag> [bench args...]
set -uo pipefail
TAG="${1:-r01}"; shift || true
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/prof_$TAG"   # (clear the local copy too before a re-run: gpurun merges, it does not mirror)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-tail $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$R/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$R/bench.py" $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$R/bench.py" $ARGS > "$OUT/bench_write.log" 2>&1
python "$R/scripts/summarize_profiles.py" "$OUT" "$TAG" > "$OUT/summary_$TAG.md" 2>&1
tail -40 "$OUT/summary_$TAG.md"
