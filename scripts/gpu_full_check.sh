#!/bin/bash
# Full GPU check of a working tree: the -m gpu suite, smoke(), one default bench line.  Run through gpurun.
mkdir -p gpurun_out
export PYTHONPATH=.
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/t_all.log 2>&1; tail -8 gpurun_out/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py > gpurun_out/bench_f16.json 2> gpurun_out/bench_f16.err; python -c "
import json; d=json.load(open('gpurun_out/bench_f16.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
