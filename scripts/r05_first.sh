#!/bin/bash
# round 5, first GPU call: the -m gpu suite on the ADVICE fixes + new parity tests, the streaming-ceiling probe, the PMC calibration
R="${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p $R/gpurun_out/r05a
O=$R/gpurun_out/r05a
export PYTHONPATH=$R
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -15 $O/tests.log
$R/scripts/micro/_bin/stream_ceiling time > $O/stream_ceiling.jsonl 2> $O/stream_ceiling.err; tail -3 $O/stream_ceiling.err
cd /tmp && export TMPDIR=/tmp
$R/scripts/micro/_bin/stream_ceiling calib > $O/calib_stdout.jsonl 2> $O/calib.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $R/scripts/micro/_bin/stream_ceiling calib > $O/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $R/scripts/micro/_bin/stream_ceiling calib > $O/calib_write.log 2>&1
python $R/scripts/micro/calib_report.py $O > $O/pmc_calibration.md 2> $O/calib_report.err; cat $O/pmc_calibration.md | tail -12
cd $R
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline_step']['frac'], d['host_busy_ms_per_step'])"
