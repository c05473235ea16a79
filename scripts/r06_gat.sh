#!/bin/bash
mkdir -p gpurun_out/r06u
export PYTHONPATH=.
timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_step_path_table.py tests/test_models_gpu.py -q -m gpu -x -k "gat or sparse_top or benchmark_scale or golden" > gpurun_out/r06u/t_gat.log 2>&1; tail -3 gpurun_out/r06u/t_gat.log
