#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r05k; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -6 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
