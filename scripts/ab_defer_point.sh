#!/usr/bin/env bash
# where the deferred sampler prefetch is launched (ops.DEFER_POINT): products benchmark, same box
for rep in 1 2; do for pt in body fwd agg none; do
if [ $pt = none ]; then export SHADOW_DEFER_PREFETCH=0; else export SHADOW_DEFER_PREFETCH=1; fi
SHADOW_DEFER_POINT=$pt timeout 300 python bench.py --no-cpu-baseline --no-tail 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('point=$pt', d['ms_per_step'], d['roofline']['frac'], 'sampler', k['sg_sample_pipeline']['avg_ms'], 'gather', k['gather_F100']['avg_ms'], 'spmm100', k['spmm_F100']['avg_ms'], 'spmm256', k['spmm_F256']['avg_ms'], 'anb', k['act_norm_bwd_nb2_F256']['avg_ms'])"
done; done
