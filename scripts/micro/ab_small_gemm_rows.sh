#!/usr/bin/env bash
# host-bound small batches: rocBLAS (>= 1024 rows only on our kernels) against our kernels from 32 rows up
for w in "--workload arxiv-khop-sage5" "--workload arxiv-khop-gcn3" "--batch 128" "--workload papers100M-ppr-sage5"; do
  for r in 1024 32 1024 32; do
    SHADOW_GEMM_SPLIT_MIN_ROWS=$r timeout 300 python bench.py $w --steps 40 --warmup 10 --no-cpu-baseline --no-tail 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w rows>=$r', d['ms_per_step'], 'host', d.get('host_busy_ms_per_step'))"
  done
done
