#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r05g; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python bench.py --workload products-khop3-gat5 --no-cpu-baseline > $O/bench_gat.json 2> $O/bench_gat.err; python -c "
import json; d=json.load(open('$O/bench_gat.json')); print(d['ms_per_step'], d['value'], d['sampler_alone']['avg_ms'], d['sampler_alone']['frac']); 
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_ms'])[:8]: print(k, v['avg_ms'], v['frac'])"
timeout 600 python bench.py --batch 512 --no-cpu-baseline --no-tail > $O/bench_b512.json 2> $O/bench_b512.err; python -c "
import json; d=json.load(open('$O/bench_b512.json')); print(d['ms_per_step'], d['value']); 
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_ms'])[:8]: print(k, v['avg_ms'], v['launches'])"
timeout 600 python bench.py --no-cpu-baseline --no-tail > $O/bench_b1024.json 2> $O/bench_b1024.err; python -c "
import json; d=json.load(open('$O/bench_b1024.json')); print(d['ms_per_step'], d['value']); 
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_ms'])[:8]: print(k, v['avg_ms'], v['launches'])"
