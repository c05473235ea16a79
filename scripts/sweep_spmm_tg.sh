#!/usr/bin/env bash
# tile-group size of the block-diagonal SpMM (SHADOW_SPMM_TG) on the k-hop and the PPR benchmark batches
for w in products-ppr-sage5 products-khop-sage5; do for tg in 0 1 2 4 8; do
  if [ $tg = 0 ]; then unset SHADOW_SPMM_TG; else export SHADOW_SPMM_TG=$tg; fi
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-tail 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$w tg=$tg', d['ms_per_step'], {n:(v['launches'], round(v['avg_ms'],4)) for n,v in k.items() if n.startswith('spmm')})"
done; done
