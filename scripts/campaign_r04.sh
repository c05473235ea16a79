cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/full_gpu_tests_r04.log
bash scripts/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1
cd $GRAFT_REPO_ROOT
bash scripts/bench_all_workloads.sh r04 > gpurun_out/bench_all_r04.log 2>&1
bash scripts/collect_workload_stats.sh r04 > gpurun_out/wstats_r04.log 2>&1
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/full_gpu_tests_r04.log; tail -12 gpurun_out/bench_all_r04.log
