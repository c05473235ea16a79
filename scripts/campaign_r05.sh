#!/bin/bash
# round 5 evidence campaign (one gpurun call): -m gpu suite, rocprofv3 trace + PMC passes of the default bench, one bench line per
# BASELINE configuration, kernel statistics of the other workloads, streaming ceiling + PMC calibration, papers100M CPU baseline
R="${GRAFT_REPO_ROOT:-$PWD}"
cd $R; export PYTHONPATH=$R
mkdir -p gpurun_out/camp_r05
[ -n "$SKIP_TESTS" ] || python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/camp_r05/full_gpu_tests.log
bash scripts/collect_profiles.sh r05 > gpurun_out/camp_r05/collect.log 2>&1
cd $R
bash scripts/bench_all_workloads.sh r05 > gpurun_out/camp_r05/bench_all.log 2>&1
bash scripts/collect_workload_stats.sh r05 > gpurun_out/camp_r05/wstats.log 2>&1
cd $R
O=$R/gpurun_out/camp_r05
$R/scripts/micro/_bin/stream_ceiling time > $O/stream_ceiling.jsonl 2> $O/stream_ceiling.err
cd /tmp && export TMPDIR=/tmp
$R/scripts/micro/_bin/stream_ceiling calib > $O/calib_stdout.jsonl 2> $O/calib.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $R/scripts/micro/_bin/stream_ceiling calib > $O/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $R/scripts/micro/_bin/stream_ceiling calib > $O/calib_write.log 2>&1
python $R/scripts/micro/calib_report.py $O > $O/pmc_calibration.md 2> $O/calib_report.err
cd $R
timeout 1500 python bench.py --workload papers100M-ppr-sage5 --steps 20 --warmup 5 --cpu-baseline-large --no-tail > $O/papers_cpu.json 2> $O/papers_cpu.err
# host side of the headline step: wall time per Python entry point / C-ABI entry, and the same-box A/B of the three forms of the row-sparse pass
python scripts/host_breakdown.py --steps 100 --warmup 10 --no-cpu-baseline --no-tail > /dev/null 2> $O/host_breakdown.txt
bash scripts/ab_top_stack.sh > $O/ab_top_stack.txt 2>&1
bash scripts/ab_root_gemm.sh products-khop3-gat5 > $O/ab_root_gemm_gat.txt 2>&1
tail -5 $O/full_gpu_tests.log; tail -12 $O/bench_all.log; tail -c 600 $O/papers_cpu.json
