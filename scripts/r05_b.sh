#!/bin/bash
# round 5, GPU call b: re-run the suite after the grad-mode fix; GEMM row-count sweep (round quantisation); depth-3 sampler phases
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r05b; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -12 $O/tests.log
for M in 262144 289309 327680 393216; do echo "M=$M"; M=$M timeout 300 python scripts/probe_gemm_fused.py 2>&1 | tail -1; done > $O/gemm_rows.log 2>&1; cat $O/gemm_rows.log
for cfg in "3 1" "3 0" "2 1" "2 0"; do set -- $cfg; echo "== depth $1 self $2"; DEPTH=$1 SELF=$2 timeout 300 python scripts/probe_sampler_batch.py 256 1024 2>&1 | tail -8; done > $O/sampler_phases.log 2>&1; cat $O/sampler_phases.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline_step']['frac'], d['host_busy_ms_per_step'])"
