#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06d; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -8 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
G="--workload products-khop3-gat5 --steps 20 --warmup 5 --no-cpu-baseline --no-tail"
for rep in 1 2; do python bench.py $G > $O/gat_new_$rep.json 2> $O/gat_new_$rep.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06d/gat_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    K = d["instrumented_steps"]
    ks = {k.replace("_F256_H4","").replace("_N256",""): v["avg_ms"] for k, v in d["kernels"].items() if v["total_ms"] / K > 0.15}
    print(f.split("/")[-1], d["ms_per_step"], "host", d["host_busy_ms_per_step"], "kern", d["roofline_step"]["kernel_ms_per_step"], d["roofline_step"]["frac"], ks)
PY
bash scripts/dist_8proc_one_gpu.sh > $O/dist8.log 2>&1; tail -c 2500 $O/dist8.log
