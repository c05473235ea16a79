#!/usr/bin/env bash
# where in the step the prefetched sampler call is issued (ops.DEFER_POINT), step time of a workload
#   usage: scripts/ab_defer_point.sh <workload> [steps]
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; export PYTHONPATH=$R
W="${1:-products-khop3-gat5}"; K="${2:-32}"
for rep in 1 2; do
 for pt in body agg fwd; do
  python bench.py --workload $W --steps $K --warmup 8 --no-cpu-baseline --no-tail --no-other-workloads --set "ops.DEFER_POINT='$pt'" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$W defer=$pt', d['ms_per_step'], 'sampler in step', d['kernels']['sg_sample_pipeline']['avg_ms'])
"
 done
 SHADOW_DEFER_PREFETCH=0 python bench.py --workload $W --steps $K --warmup 8 --no-cpu-baseline --no-tail --no-other-workloads 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$W immediate', d['ms_per_step'], 'sampler in step', d['kernels']['sg_sample_pipeline']['avg_ms'])
"
done
