#!/bin/bash
# Sampler tests + depth-3 / depth-2 call timing.
mkdir -p gpurun_out/r06t
export PYTHONPATH=.
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_ref_configs_gpu.py -q -m gpu -x > gpurun_out/r06t/t_samp.log 2>&1; tail -3 gpurun_out/r06t/t_samp.log
DEPTH=3 SELF=1 timeout 200 python scripts/probe_sampler_batch.py 256 2>&1 | grep -v amdgpu.ids | grep "B=\|phases\|wave 0"
DEPTH=2 SELF=0 timeout 200 python scripts/probe_sampler_batch.py 1024 4096 2>&1 | grep -v amdgpu.ids | grep "B=\|phases\|wave 0"
