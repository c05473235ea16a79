#!/bin/bash
# GAT tests + configs[3] bench A/B (mapped row gradient on / off).
mkdir -p gpurun_out/r06u
export PYTHONPATH=.
timeout 1200 python -m pytest tests/test_layers_gpu.py -q -m gpu -x -k "gat or sparse_top or benchmark_scale" > gpurun_out/r06u/t_gat.log 2>&1; tail -5 gpurun_out/r06u/t_gat.log
for v in True False True False; do
  timeout 300 python bench.py --workload products-khop3-gat5 --steps 20 --warmup 5 --no-cpu-baseline --no-tail --no-other-workloads --set ops_gat.MAP_ROWS_GRADIENT=$v > gpurun_out/r06u/gat_map_$v.json 2> gpurun_out/r06u/gat_map_$v.err
  python - gpurun_out/r06u/gat_map_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
ks = {k["name"]: k for k in d["kernels"]} if isinstance(d.get("kernels"), list) else d.get("kernels", {})
pick = {n: round(v.get("avg_us", v.get("us", 0)), 1) for n, v in ks.items() if n.startswith(("gat_bwd", "act_norm_bwd"))} if isinstance(ks, dict) else {}
print("MAP", sys.argv[2], d["ms_per_step"], d["roofline_step"]["kernel_ms_per_step"], d["host_busy_ms_per_step"], pick)
PY
done
