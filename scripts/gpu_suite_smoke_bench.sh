#!/bin/bash
# GPU box: the whole -m gpu suite, smoke(), then the default bench command timed (what the driver runs at round end)
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/suite; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -4 $O/tests.log; grep -n "^E " $O/tests.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; grep "other_workloads\|real" $O/bench_default.err | cut -c1-400; head -c 400 $O/bench_default.json
