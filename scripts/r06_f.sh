#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06f; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
bash scripts/bench_all_workloads.sh r06 > $O/bench_all.log 2>&1; tail -12 $O/bench_all.log
bash scripts/collect_profiles.sh r06 > $O/collect.log 2>&1; tail -5 $O/collect.log
bash scripts/collect_workload_stats.sh r06 > $O/wstats.log 2>&1; tail -5 $O/wstats.log
bash scripts/pmc_workload.sh gat_after products-khop3-gat5 > $O/pmc_gat.log 2>&1; tail -3 $O/pmc_gat.log
bash scripts/pmc_workload.sh ppr products-ppr-sage5 > $O/pmc_ppr.log 2>&1; tail -3 $O/pmc_ppr.log
bash scripts/pmc_workload.sh arxiv arxiv-khop-sage5 > $O/pmc_arxiv.log 2>&1; tail -3 $O/pmc_arxiv.log
