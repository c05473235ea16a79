#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06l; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
run() { (cd $2 && PYTHONPATH=$PWD python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-tail) > $O/$1.json 2> $O/$1.err; }
for rep in 1 2; do
  run head_pre_$rep _ab/prefill ""
  run head_new_$rep . ""
  run gat_pre_$rep _ab/prefill "--workload products-khop3-gat5"
  run gat_new_$rep . "--workload products-khop3-gat5"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06l/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    print(f.split("/")[-1], d["ms_per_step"], "host", d["host_busy_ms_per_step"], "kern", d["roofline_step"]["kernel_ms_per_step"], d["roofline_step"]["frac"])
PY
timeout 900 python -m pytest tests/test_layers_gpu.py -q -m gpu -x -k "sparse_top or benchmark_scale or timed_configuration" 2>&1 | tail -3
