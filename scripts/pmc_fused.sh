#!/usr/bin/env bash
# SQ counters of the two GEMM-epilogue kernels at the benchmark's shape (separate PMC passes, kernel-trace only):
# how busy the VALU, the matrix pipe, LDS and the vector-memory path are, and how much of the wave time is waiting.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/pmc_fused"; rm -rf "$OUT"; mkdir -p "$OUT"
export PYTHONPATH="$R"
cd /tmp && export TMPDIR=/tmp
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-60)
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$tag" -- python "$R/scripts/probe_fused_pair.py" > "$OUT/$tag.log" 2>&1
  find "$OUT/$tag" -name "*kernel_trace.csv" -delete
done
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "gemm_nt_fused_kernel" not in kn: continue
        k = (kn[kn.index("gemm_nt_fused_kernel"):][:44], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for (kn, cn), (v, c) in sorted(acc.items()):
        print(f"{kn:44s} {cn:30s} per-launch {v / c:16.0f}  (launches {c})")
PY
