#!/bin/bash
mkdir -p gpurun_out/r06u
R=$PWD
for t in "$R" "$R/_ab/base"; do
  cd $t; PYTHONPATH=$t python bench.py --workload products-khop3-gat5 --steps 20 --warmup 8 --no-cpu-baseline --no-tail --no-other-workloads 2>/dev/null > $R/gpurun_out/r06u/k_$(basename $t).json
done
