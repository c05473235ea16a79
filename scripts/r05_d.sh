#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r05d; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 1500 python -m pytest tests/test_sampler_gpu.py tests/test_minibatch_gpu.py -q -m gpu -x > $O/tests.log 2>&1; tail -5 $O/tests.log
for cfg in "3 1" "2 1"; do set -- $cfg; echo "== depth $1 self $2"; DEPTH=$1 SELF=$2 timeout 300 python scripts/probe_sampler_batch.py 256 1024 2>&1 | tail -8; done > $O/sampler_phases.log 2>&1; cat $O/sampler_phases.log
echo "== depth 3 self 1, 64 KB filter, capm 3072"; SHADOW_SG_BITWORDS=16384 SHADOW_SG_CAPM=3072 DEPTH=3 SELF=1 timeout 300 python scripts/probe_sampler_batch.py 256 2>&1 | tail -4
echo "== depth 3 self 1, 64 KB filter, capm 2048"; SHADOW_SG_BITWORDS=16384 SHADOW_SG_CAPM=2048 DEPTH=3 SELF=1 timeout 300 python scripts/probe_sampler_batch.py 256 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
DEPTH=3 SELF=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/scripts/probe_sampler_batch.py 256 > $O/trace.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sg_" in r["Name"]: print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, "us")
PY
