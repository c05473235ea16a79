#!/usr/bin/env bash
# Same-box A/B: the row-sparse backward's small products (over the roots / the rows T) on the library's kernels (own) or through
# torch.mm / rocBLAS (torch), for one workload.   usage: scripts/ab_root_gemm.sh <workload>
wl=${1:-products-khop3-gat5}
out=gpurun_out/ab_root_gemm; mkdir -p $out
for rep in 1 2; do
  for v in own torch; do
    timeout 300 python -c "
import sys, runpy
import shadow_gnn_amd.ops as o
o.ROOT_GEMM_MIN_ROWS = 64 if '$v' == 'own' else 1 << 30
sys.argv = ['bench.py', '--workload', '$wl', '--steps', '40', '--warmup', '8', '--no-cpu-baseline', '--no-tail']
runpy.run_path('bench.py', run_name='__main__')" > $out/${wl}_${v}_$rep.json 2> $out/${wl}_${v}_$rep.err
    python - <<PY
import json
d = json.loads(open('$out/${wl}_${v}_$rep.json').read().strip().splitlines()[-1])
print('$wl $v rep $rep: ms/step', d['ms_per_step'], 'host_busy', d['host_busy_ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'])
PY
  done
done
