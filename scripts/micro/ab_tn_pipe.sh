#!/usr/bin/env bash
# same-box A/B of the headline step: interleaved weight-gradient step and line tiles of the layer-0 SpMM on / off
for v in "0 0" "1 1" "0 0" "1 1"; do set -- $v
  SHADOW_GEMM_TN_PIPE=$1 SHADOW_SPMM_LINES=$2 python bench.py --no-cpu-baseline --no-tail 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('pipe=$1 lines=$2', d['ms_per_step'], {n:round(v['avg_ms'],4) for n,v in k.items() if 'tn_split_N256' == n[5:] or n=='spmm_F100'})"
done
