#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06a; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 1200 python -m pytest tests/test_sampler_gpu.py tests/test_minibatch_gpu.py -q -m gpu -x -k "full_size or configs3 or eight_rank or three_rank" > $O/tests.log 2>&1; tail -5 $O/tests.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.err; head -c 600 $O/bench_default.json
bash scripts/dist_8proc_one_gpu.sh > $O/dist8.log 2>&1; tail -c 1500 $O/dist8.log
bash scripts/pmc_workload.sh gat products-khop3-gat5 > $O/pmc_gat.log 2>&1; tail -30 $O/pmc_gat.log
