#!/bin/bash
# Round-6 measurement campaign on the final tree (run through scripts/run_campaign.sh r06): the -m gpu suite, one bench line per
# BASELINE configuration, rocprofv3 kernel statistics + FETCH / WRITE counter passes of the headline, kernel statistics of the other
# workloads, the counter passes of configs[3], the depth-3 sampler call's phases.
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06f; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $R/gpurun_out/full_gpu_tests_r06.log 2>&1; tail -3 $R/gpurun_out/full_gpu_tests_r06.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/bench_all_workloads.sh r06 > $O/bench_all.log 2>&1; tail -12 $O/bench_all.log
bash scripts/collect_profiles.sh r06 > $O/collect.log 2>&1; tail -5 $O/collect.log
bash scripts/collect_workload_stats.sh r06 > $O/wstats.log 2>&1; tail -5 $O/wstats.log
bash scripts/pmc_workload.sh gat_after products-khop3-gat5 > $O/pmc_gat.log 2>&1; tail -3 $O/pmc_gat.log
for cfg in "3 1" "3 0" "2 1"; do set -- $cfg; echo "== depth $1 self $2"; DEPTH=$1 SELF=$2 python scripts/probe_sampler_batch.py 256 1024 2>&1 | grep -v amdgpu.ids; done > $O/sampler_phases.txt 2>&1
