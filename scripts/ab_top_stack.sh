#!/usr/bin/env bash
# Same-box A/B of the headline step's host side:
#   stack      row-sparse top pass issued from the whole-stack node, the roots' products on the library's kernels (default)
#   torchmm    ... the roots' products through torch.mm / rocBLAS (round 4's form)
#   layers     row-sparse top pass from the layer-by-layer nodes (round 4's autograd graph), library kernels
out=gpurun_out/ab_top_stack; mkdir -p $out
for rep in 1 2; do
  for v in stack torchmm layers; do
    timeout 300 python -c "
import sys, runpy
import shadow_gnn_amd.ops as o
if '$v' == 'layers': o.SPARSE_TOP_STACK = False
if '$v' == 'torchmm': o.ROOT_GEMM_MIN_ROWS = 1 << 30
sys.argv = ['bench.py', '--steps', '60', '--warmup', '10', '--no-cpu-baseline', '--no-tail']
runpy.run_path('bench.py', run_name='__main__')" > $out/${v}_$rep.json 2> $out/${v}_$rep.err
    python - <<PY
import json
d = json.loads(open('$out/${v}_$rep.json').read().strip().splitlines()[-1])
print('$v rep $rep: ms/step', d['ms_per_step'], 'host_busy', d['host_busy_ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'])
PY
  done
done
