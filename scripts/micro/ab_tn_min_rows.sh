#!/bin/bash
# rows per workgroup of the weight-gradient kernel at small batches (arxiv SAGE-5: M = 39.5 k)
export PYTHONPATH=.
for r in 256 128 64 256 128 64; do echo "min_rows=$r"; SHADOW_GEMM_TN_MIN_ROWS=$r python scripts/probe_gemm_tn.py 2>&1 | grep "M=39500 K=256 round 1\|M=289309 K=256 round 1"; done
for r in 256 128 64 256 128; do SHADOW_GEMM_TN_MIN_ROWS=$r python bench.py --workload arxiv-khop-sage5 --steps 40 --warmup 10 --no-cpu-baseline --no-tail 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('arxiv-sage5 min_rows=$r', d['ms_per_step'], {n:round(v['avg_ms'],4) for n,v in k.items() if 'tn_split' in n})"; done
