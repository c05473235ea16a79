#!/usr/bin/env bash
# Same-box A/B: the paired weight-gradient kernel on half the row slices (one round of workgroups, half the partial products).
for rep in 1 2; do for v in 1 0; do
  SHADOW_GEMM_TN_PAIR_HALF=$v timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-tail > /tmp/ab.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print('pair_half=$v rep $rep: ms/step', d['ms_per_step'], ' gemm_tn_f16_pair', d['kernels']['gemm_tn_f16_pair_N256']['avg_ms'])
PY
done; done
