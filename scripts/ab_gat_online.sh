#!/usr/bin/env bash
# A/B of the one-pass (online) softmax of gat_row_fwd_kernel (csrc/gat.hip, SHADOW_GAT_ONLINE_SOFTMAX = 1 | 0).
run() {
  python bench.py --workload products-khop3-gat5 --steps 30 --warmup 6 --no-cpu-baseline --no-tail 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$1', d['ms_per_step'], {n:round(k[n]['avg_ms']*1e3,1) for n in k if n.startswith('gat_')})"
}
for m in 1 0 1 0; do
  touch shadow_gnn_amd/csrc/gat.hip
  SHADOW_HIPCC_FLAGS="-DSHADOW_GAT_ONLINE_SOFTMAX=$m" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  run "ONLINE=$m"
done
