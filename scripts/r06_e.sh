#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06e; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python -m pytest tests/test_layers_gpu.py -q -m gpu -k "paired_linear_equal or act_norm_tail_equals or one_edge_walk" > $O/tests.log 2>&1; tail -5 $O/tests.log
grep -n "^E " $O/tests.log | head -40
