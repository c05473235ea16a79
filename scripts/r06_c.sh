#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06c; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_layers_oracle_golden.py tests/test_ref_configs_gpu.py tests/test_tail_gpu.py -q -m gpu -x -k "gat or GAT or path_cell or step_path or golden or benchmark_scale" > $O/tests_gat.log 2>&1; tail -8 $O/tests_gat.log
G="--workload products-khop3-gat5 --steps 20 --warmup 5 --no-cpu-baseline --no-tail"
run() { # name dir extra
  (cd $2 && PYTHONPATH=$PWD python bench.py $G $3) > $O/gat_$1.json 2> $O/gat_$1.err
}
for rep in 1 2; do
  run r05_$rep _ab/r05 ""
  run new_$rep . ""
  run unfused_$rep . "--set ops_gat.FUSED_FWD_TAIL=False"
  run colw6_$rep _ab/colw6 ""
  run colw7_$rep _ab/colw7 ""
  run zsearly_$rep _ab/zsearly ""
  run libm_$rep _ab/libm ""
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06c/gat_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    K = d["instrumented_steps"]
    ks = {k.replace("_F256_H4","").replace("_N256",""): v["avg_ms"] for k, v in d["kernels"].items() if v["total_ms"] / K > 0.15}
    print(f.split("/")[-1], d["ms_per_step"], "host", d["host_busy_ms_per_step"], "kern", d["roofline_step"]["kernel_ms_per_step"], ks)
PY
for NR in 8; do
  SHADOW_DIST_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $NR --master-addr 127.0.0.1 --master-port 2961$NR \
      bench.py --gpus $NR --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-tail --no-other-workloads --hang-dump-after 200 > $O/dist_$NR.json 2> $O/dist_$NR.err
  echo "ranks $NR rc=$? $(grep '^{' $O/dist_$NR.json | head -c 400)"
done
