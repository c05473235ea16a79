#!/usr/bin/env bash
# Instruction-mix counters of the sampler kernels (separate PMC passes, kernel-trace only).
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/pmc_sampler"; rm -rf "$OUT"; mkdir -p "$OUT"
export PYTHONPATH="$R"
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$tag" -- python "$R/scripts/probe_sampler.py" --iters 5 "$@" > "$OUT/$tag.log" 2>&1
done
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for (kn, cn), (v, c) in sorted(acc.items()):
        if "sg_" in kn:
            print(f"{kn:60s} {cn:24s} per-launch {v / c:14.0f}  (launches {c})")
PY
