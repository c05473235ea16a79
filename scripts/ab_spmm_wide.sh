#!/usr/bin/env bash
# Same-box A/B of the step: aggregations of 256-float rows by the row bound (1, default) / always the block-diagonal LDS kernel (0) / always the pipelined CSR kernel (2).
out=gpurun_out/ab_spmm_wide; mkdir -p $out
for wl in ${WORKLOADS:-products-khop-sage5 products-ppr-sage5 arxiv-khop-sage5}; do
for rep in 1 2; do
  for v in 1 0 2; do
    timeout 300 python -c "
import sys, runpy
from shadow_gnn_amd import _lib
_lib.load().sl_set_spmm_wide_pipe($v)
sys.argv = ['bench.py', '--workload', '$wl', '--steps', '40', '--warmup', '8', '--no-cpu-baseline', '--no-tail']
runpy.run_path('bench.py', run_name='__main__')" > $out/${wl}_${v}_$rep.json 2> $out/${wl}_${v}_$rep.err
    python - <<PY
import json
d = json.loads(open('$out/${wl}_${v}_$rep.json').read().strip().splitlines()[-1])
k = d['kernels']
print('$wl wide_pipe=$v rep $rep: ms/step', d['ms_per_step'], ' spmm_F256', k.get('spmm_F256', {}).get('avg_ms'), ' host_busy', d['host_busy_ms_per_step'])
PY
  done
done
done
