#!/usr/bin/env bash
# Does the sampler call's scan kernel leave room for the training stream?  Its 16-wavefront workgroups at 126 VGPRs fill a CU's
# register file; fewer wavefronts per workgroup (SHADOW_SG_SCAN_THREADS) / a normal-priority prefetch stream, step time of a workload.
#   usage: scripts/ab_scan_share.sh <workload> [steps]
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; export PYTHONPATH=$R
W="${1:-products-khop3-gat5}"; K="${2:-32}"
for rep in 1 2; do
 for cfg in "0 -1" "512 -1" "256 -1" "0 0" "512 0"; do set -- $cfg
  SHADOW_SG_SCAN_THREADS=$1 SHADOW_PREFETCH_PRIORITY=$2 python bench.py --workload $W --steps $K --warmup 8 --no-cpu-baseline --no-tail --no-other-workloads 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$W threads=$1 prio=$2', d['ms_per_step'], 'sampler alone', d['sampler_alone']['avg_ms'], 'in step', d['kernels']['sg_sample_pipeline']['avg_ms'])
"
 done
done
