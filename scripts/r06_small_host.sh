#!/bin/bash
# Host-side profile of the small-batch workloads (reference batch sizes) + one headline line.
mkdir -p gpurun_out/r06s
export PYTHONPATH=.
timeout 200 python scripts/profile_host.py sage > gpurun_out/r06s/prof_sage.txt 2>&1; head -3 gpurun_out/r06s/prof_sage.txt
timeout 200 python scripts/profile_host.py gcn > gpurun_out/r06s/prof_gcn.txt 2>&1; head -3 gpurun_out/r06s/prof_gcn.txt
for spec in "products-khop-sage5 128" "arxiv-khop-gcn3 0"; do
  set -- $spec
  timeout 240 python scripts/host_breakdown.py --workload $1 --batch $2 --steps 100 --warmup 10 --no-cpu-baseline --no-tail --no-other-workloads \
     > gpurun_out/r06s/hb_$1_$2.json 2> gpurun_out/r06s/hb_$1_$2.txt
  tail -3 gpurun_out/r06s/hb_$1_$2.txt
done
timeout 300 python bench.py --no-other-workloads --no-cpu-baseline > gpurun_out/r06s/bench_default.json 2> gpurun_out/r06s/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r06s/bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['host_busy_ms_per_step'])"
