#!/usr/bin/env bash
# SQ counters of the split-bf16 GEMM kernels (separate PMC passes, kernel-trace only).
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/pmc_gemm"; rm -rf "$OUT"; mkdir -p "$OUT"
export PYTHONPATH="$R"
cd /tmp && export TMPDIR=/tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$tag" -- python "$R/scripts/probe_gemm_split.py" > "$OUT/$tag.log" 2>&1
done
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for (kn, cn), (v, c) in sorted(acc.items()):
        if "gemm_nt_split_kernel<1, 8, false>" in kn or "gemm_tn_split_kernel<4>" in kn:
            print(f"{kn:48s} {cn:30s} per-launch {v / c:16.0f}  (launches {c})")
PY
