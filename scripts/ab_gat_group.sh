#!/usr/bin/env bash
# A/B of the edge-group sizes of the GAT edge kernels (csrc/gat.hip, SHADOW_GAT_GROUPS_{FWD,ROW,COL} = bit masks of {8, 4, 2}): the
# library is rebuilt on the box with each setting (gat.hip only), then the GAT benchmark's step time and kernel classes.
run() {
  python bench.py --workload products-khop3-gat5 --steps 30 --warmup 6 --no-cpu-baseline --no-tail 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$1', d['ms_per_step'], {n:round(k[n]['avg_ms']*1e3,1) for n in k if n.startswith('gat_')})"
}
for m in "6 2 2" "6 2 6" "6 2 4" "6 0 2" "6 4 2" "6 2 0" "6 2 2"; do
  set -- $m
  touch shadow_gnn_amd/csrc/gat.hip
  SHADOW_HIPCC_FLAGS="-DSHADOW_GAT_GROUPS_FWD=$1 -DSHADOW_GAT_GROUPS_ROW=$2 -DSHADOW_GAT_GROUPS_COL=$3" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  run "FWD=$1 ROW=$2 COL=$3"
done
