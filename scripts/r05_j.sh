#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r05j; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 2400 python -m pytest tests/test_layers_gpu.py tests/test_tail_gpu.py tests/test_ref_configs_gpu.py -q -m gpu -x -k "gat or GAT or golden or ref_config or fuzz" > $O/tests.log 2>&1; tail -4 $O/tests.log
for hn in 1 0; do
python -c "
import sys, runpy
import shadow_gnn_amd.ops_gat as g
g.RECOMPUTE_HN = bool($hn)
sys.argv = ['bench.py', '--workload', 'products-khop3-gat5', '--no-cpu-baseline', '--no-tail']
runpy.run_path('bench.py', run_name='__main__')" > $O/bench_gat_hn$hn.json 2> $O/bench_gat_hn$hn.err
python -c "
import json; d=json.load(open('$O/bench_gat_hn$hn.json')); print('recompute_hn=$hn', d['ms_per_step'], d['value']);
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_ms'])[:4]: print(k, v['avg_ms'], v['frac'])"
done
