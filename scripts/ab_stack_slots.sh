#!/usr/bin/env bash
# Same-box A/B: the whole-stack node's saved forward products as ONE [4 L - 1, n, F] tensor (0) or one allocation each (1), beside the
# layer-by-layer nodes (layers).
for rep in 1 2; do for v in 0 1 layers; do
  if [ $v = layers ]; then
    timeout 300 python -c "
import sys, runpy
import shadow_gnn_amd.ops as o
o.SPARSE_TOP_STACK = False
sys.argv = ['bench.py', '--steps', '60', '--warmup', '10', '--no-cpu-baseline', '--no-tail']
runpy.run_path('bench.py', run_name='__main__')" > /tmp/ab.json 2>/dev/null
  else
    SHADOW_STACK_SEPARATE_SLOTS=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-tail > /tmp/ab.json 2>/dev/null
  fi
  python - <<PY
import json
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print('slots=$v rep $rep: ms/step', d['ms_per_step'], ' kernel ms', d['roofline_step']['kernel_ms_per_step'], ' host_busy', d['host_busy_ms_per_step'], d['allocator_in_timed_region'])
PY
done; done
