#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06k; mkdir -p $O
cd $R/_ab/timing
export PYTHONPATH=$PWD
for cfg in "3 1" "3 0" "2 1"; do set -- $cfg; echo "== depth $1 self $2"; DEPTH=$1 SELF=$2 python probe_sampler_batch.py 256 1024 2>&1 | grep -v amdgpu.ids; done | tee $O/sampler_phases.txt
cd $R; export PYTHONPATH=$R
DEPTH=3 SELF=1 bash scripts/prof_sampler.sh 256 2>&1 | tail -12 | tee $O/sampler_kernels.txt
