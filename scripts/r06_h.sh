#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06h; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests/test_layers_gpu.py -q -m gpu -x -k "dual or ppr_mean_pool or step_path or fused_output_dropout or readout or chained" > $O/tests_sel.log 2>&1; tail -6 $O/tests_sel.log; grep -n "^E " $O/tests_sel.log | head -20
run() { (cd $2 && PYTHONPATH=$PWD python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-tail) > $O/$1.json 2> $O/$1.err; }
for rep in 1 2; do
  run ppr_new_$rep . "--workload products-ppr-sage5"
  run ppr_nochain_$rep . "--workload products-ppr-sage5 --set ops.CHAIN_DUAL=False"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06h/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    K = d["instrumented_steps"]
    ks = {k.replace("_F256_H4","").replace("_N256","").replace("_nb2",""): (round(v["launches"]/K,1), v["avg_ms"]) for k, v in d["kernels"].items() if v["total_ms"] / K > 0.12}
    print(f.split("/")[-1], d["ms_per_step"], "host", d["host_busy_ms_per_step"], "kern", d["roofline_step"]["kernel_ms_per_step"], d["roofline_step"]["frac"], ks)
PY
