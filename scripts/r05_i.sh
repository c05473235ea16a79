#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r05i; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 2400 python -m pytest tests/test_layers_gpu.py tests/test_tail_gpu.py tests/test_minibatch_gpu.py -q -m gpu -x -k "sparse or top or tail or plan or timed or benchmark_scale or epoch or reproducible" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-tail > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d['roofline_step']['frac'], d['host_busy_ms_per_step']); 
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_ms'])[:10]: print(k, v['avg_ms'], v['launches'], v['frac'])"
