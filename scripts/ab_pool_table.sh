#!/usr/bin/env bash
# round 6: pooled read-out gradient as a table -- tests, then the configs[2] step with the table on / off (same box)
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; export PYTHONPATH=$R
O=$R/gpurun_out/r06o; mkdir -p $O
timeout 1200 python -m pytest tests/test_layers_gpu.py -x -q -m gpu -k "table or dual or pool" 2>&1 | tail -6 | tee $O/tests.txt
for on in 1 0 1 0; do
  python bench.py --workload products-ppr-sage5 --steps 30 --warmup 8 --no-cpu-baseline --no-other-workloads --set ops.POOL_GRAD_TABLE=$( [ $on = 1 ] && echo True || echo False ) 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('table=$on', d['ms_per_step'], 'kernel_ms', d['roofline_step']['kernel_ms_per_step'], {n: round(v['avg_ms'], 4) for n, v in k.items() if 'an_bwd' in n or 'pool' in n})
"
done | tee $O/ab.txt
