#!/bin/bash
# round 5, GPU call c: sampler tests on the reworked flat scan (row walk per round, sampled node list, self edges), phases
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r05c; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 1500 python -m pytest tests/test_sampler_gpu.py tests/test_minibatch_gpu.py -q -m gpu -x > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout 600 python -m pytest tests/test_layers_gpu.py -q -m gpu -x -k "ppr_mean_pool" > $O/tests2.log 2>&1; tail -4 $O/tests2.log
for cfg in "3 1" "3 0" "2 1" "2 0"; do set -- $cfg; echo "== depth $1 self $2"; DEPTH=$1 SELF=$2 timeout 300 python scripts/probe_sampler_batch.py 256 1024 2>&1 | tail -8; done > $O/sampler_phases.log 2>&1; cat $O/sampler_phases.log
echo "== depth 3 self 1, 64 KB filter"; SHADOW_SG_BITWORDS=16384 DEPTH=3 SELF=1 timeout 300 python scripts/probe_sampler_batch.py 256 2>&1 | tail -4
