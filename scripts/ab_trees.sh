#!/usr/bin/env bash
# same-box A/B of one workload: this tree against an exported tree under _ab/<name> (git archive <rev> | tar -x -C _ab/<name>; built there)
#   usage: scripts/ab_trees.sh <name> <workload> [steps]
R="${GRAFT_REPO_ROOT:-$PWD}"; N="${1:?tree}"; W="${2:-products-khop-sage5}"; K="${3:-30}"
for i in 1 2; do
  for t in "$R" "$R/_ab/$N"; do
    cd $t; PYTHONPATH=$t python bench.py --workload $W --steps $K --warmup 8 --no-cpu-baseline --no-tail --no-other-workloads 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t'.split('/')[-1], '$W', d['ms_per_step'], 'host_busy', d['host_busy_ms_per_step'], 'kernel_ms', d['roofline_step']['kernel_ms_per_step'])
"
  done
done
