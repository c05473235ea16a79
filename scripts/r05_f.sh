#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r05f; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python bench.py --workload products-khop3-gat5 --no-cpu-baseline > $O/bench_gat.json 2> $O/bench_gat.err; python -c "
import json; d=json.load(open('$O/bench_gat.json')); print(d['ms_per_step'], d['value'], d['sampler_alone'])"
timeout 600 python bench.py --workload arxiv-khop-gcn3 --no-cpu-baseline > $O/bench_gcn.json 2> $O/bench_gcn.err; python -c "
import json; d=json.load(open('$O/bench_gcn.json')); print(d['ms_per_step'], d['value'], d['sampler_alone'])"
