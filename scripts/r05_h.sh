#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r05h; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -5 $O/tests.log
DEPTH=3 SELF=1 timeout 300 python scripts/probe_sampler_batch.py 256 2>&1 | tail -6
timeout 600 python bench.py --no-cpu-baseline --no-tail > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d['roofline_step']['frac'], d['host_busy_ms_per_step']); 
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_ms'])[:12]: print(k, v['avg_ms'], v['launches'], v['frac'])"
