#!/usr/bin/env bash
# SQ counters of the sampler kernels (separate PMC passes, kernel-trace only) over scripts/probe_sampler_batch.py.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/pmc_sampler"; rm -rf "$OUT"; mkdir -p "$OUT"
export PYTHONPATH="$R"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -- python "$R/scripts/probe_sampler_batch.py" "${@:-1024}" > "$OUT/p$i.log" 2>&1
done
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:44], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for (kn, cn), (v, c) in sorted(acc.items()):
        if "sg_" in kn:
            print(f"{kn:44s} {cn:24s} per-launch {v / c:16.0f}  (launches {c})")
PY
