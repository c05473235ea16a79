#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06i; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_ref_configs_gpu.py tests/test_minibatch_gpu.py -q -m gpu -x -k "gat or GAT or lazy or reproducible" > $O/tests_sel.log 2>&1; tail -4 $O/tests_sel.log; grep -n "^E " $O/tests_sel.log | head -10
for rep in 1 2; do python bench.py --workload products-khop3-gat5 --steps 20 --warmup 5 --no-cpu-baseline --no-tail > $O/gat_new_$rep.json 2> $O/gat_new_$rep.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06i/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    K = d["instrumented_steps"]
    ks = {k.replace("_F256_H4","").replace("_N256","").replace("_nb2",""): (round(v["launches"]/K,1), v["avg_ms"]) for k, v in d["kernels"].items() if v["total_ms"] / K > 0.02}
    print(f.split("/")[-1], d["ms_per_step"], "host", d["host_busy_ms_per_step"], "kern", d["roofline_step"]["kernel_ms_per_step"], d["roofline_step"]["frac"], ks)
PY
