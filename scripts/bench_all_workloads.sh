#!/bin/bash
# One bench line per BASELINE.json configuration (+ the reference's own products batch size) into gpurun_out/bench_<tag>/
#   usage: scripts/bench_all_workloads.sh <tag>
TAG="${1:-r03}"
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/bench_$TAG"; mkdir -p "$OUT"
cd "$R"
python bench.py > "$OUT/default.json" 2> "$OUT/default.err"
for w in arxiv-khop-gcn3 arxiv-khop-sage5 products-ppr-sage5 products-khop3-gat5 papers100M-ppr-sage5; do
  python bench.py --workload $w --steps 20 --warmup 5 > "$OUT/$w.json" 2> "$OUT/$w.err"
done
# config_train/products/vanilla/sage_5_khop.yml:20 batch_size: 128 (the reference's own batch size for the headline model)
python bench.py --batch 128 --steps 30 --warmup 5 --no-cpu-baseline --no-tail > "$OUT/products-khop-sage5_b128.json" 2> "$OUT/products-khop-sage5_b128.err"
for f in "$OUT"/*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], 'ms/step', d['value'], d['unit'], 'host_busy', d.get('host_busy_ms_per_step'), 'roofline', d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
