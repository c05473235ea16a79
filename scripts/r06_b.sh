#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06b; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
# 1. GAT changes: the GAT / layer tests first
timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_layers_oracle_golden.py tests/test_ref_configs_gpu.py tests/test_tail_gpu.py -q -m gpu -x -k "gat or GAT or path_cell or step_path or golden or benchmark_scale" > $O/tests_gat.log 2>&1; tail -8 $O/tests_gat.log
# 2. same-box A/B: round-5 tree vs this tree on the GAT workload
for rep in 1 2; do
  (cd _ab/r05 && PYTHONPATH=$R/_ab/r05 python bench.py --workload products-khop3-gat5 --steps 20 --warmup 5 --no-cpu-baseline --no-tail) > $O/gat_r05_$rep.json 2> $O/gat_r05_$rep.err
  python bench.py --workload products-khop3-gat5 --steps 20 --warmup 5 --no-cpu-baseline --no-tail > $O/gat_new_$rep.json 2> $O/gat_new_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06b/gat_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    K = d["instrumented_steps"]
    ks = {k: (round(v["launches"] / K, 1), v["avg_ms"]) for k, v in d["kernels"].items() if v["total_ms"] / K > 0.15}
    print(f, d["ms_per_step"], "host", d["host_busy_ms_per_step"], "kern", d["roofline_step"]["kernel_ms_per_step"], ks)
PY
# 3. the 8-rank hang: 2 / 4 / 8 ranks, stacks dumped after 100 s
for NR in 2 4 8; do
  SHADOW_DIST_BACKEND=gloo timeout 260 python -m torch.distributed.run --nnodes=1 --nproc-per-node $NR --master-addr 127.0.0.1 --master-port 2961$NR \
      bench.py --gpus $NR --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-tail --no-other-workloads --hang-dump-after 100 > $O/dist_$NR.json 2> $O/dist_$NR.err
  echo "ranks $NR rc=$? $(head -c 300 $O/dist_$NR.json)"
done
grep -n "File \"/\|Thread\|most recent" $O/dist_8.err | head -80
