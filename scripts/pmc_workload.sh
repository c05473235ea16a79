#!/usr/bin/env bash
# Counter passes of ONE bench.py workload, kernel by kernel (run on the GPU box through gpurun):
#   * HBM traffic: FETCH_SIZE / WRITE_SIZE, one pass each (never combined with trace domains)
#   * SQ issue / wait counters in groups of three, one pass per group
#   * L2 hit / miss
# summarised per kernel (mean per launch) into gpurun_out/pmc_<tag>/summary.txt
#   usage: scripts/pmc_workload.sh <tag> <workload> [kernel-name filter, default 'shadow::']
set -uo pipefail
TAG="${1:?tag}"; WL="${2:?workload}"; FILT="${3:-shadow::}"
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/pmc_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
export PYTHONPATH="$R"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-tail --no-other-workloads"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  tag=$(printf "%02d_" $i)$(echo $C | tr ' ' '_' | cut -c1-60)
  SHADOW_BENCH_NO_KTIMER=1 timeout 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/$tag" -- python "$R/bench.py" $ARGS > "$OUT/$tag.log" 2>&1
  echo "[pmc] $C rc=$?"
done
python - "$OUT" "$FILT" <<'PY' | tee "$OUT/summary.txt"
import sys, glob, csv, collections, re
out, filt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if filt not in kn:
            continue
        kn = re.sub(r"\(.*", "", kn[kn.index(filt):])[:70].replace(", ", " ").replace(",", " ")
        k = (kn, r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
kern = sorted({k for k, _ in acc})
cnames = sorted({c for _, c in acc})
print("kernel,launches," + ",".join(cnames))
for k in kern:
    n = max(acc[(k, c)][1] for c in cnames if (k, c) in acc)
    print(k + f",{n}," + ",".join(f"{acc[(k, c)][0] / acc[(k, c)][1]:.0f}" if (k, c) in acc else "" for c in cnames))
PY
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
