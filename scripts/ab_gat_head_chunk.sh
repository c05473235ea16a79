#!/usr/bin/env bash
# Same-box sweep of the head-split walk's chunk height (SHADOW_GAT_HEAD_CHUNK; 0 = the whole-row kernels): gat_fwd / gat_bwd per
# launch and the step of the products depth-3 GAT workload.
out=gpurun_out/ab_gat_head_chunk; mkdir -p $out
for ch in ${CHUNKS:-0 4096 2048 8192 0 4096}; do
  SHADOW_GAT_HEAD_CHUNK=$ch timeout 300 python bench.py --workload products-khop3-gat5 --steps 30 --warmup 6 --no-cpu-baseline --no-tail > $out/ch$ch.json 2> $out/ch$ch.err
  python - <<PY
import json
d = json.loads(open('$out/ch$ch.json').read().strip().splitlines()[-1])
k = d['kernels']
print('chunk $ch: ms/step', d['ms_per_step'], ' gat_fwd', k['gat_fwd_F256_H4']['avg_ms'], ' gat_bwd', k['gat_bwd_F256_H4']['avg_ms'], ' gat_bwd_rows', k.get('gat_bwd_rows_F256_H4', {}).get('avg_ms'))
PY
done
