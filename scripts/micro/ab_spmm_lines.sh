#!/usr/bin/env bash
# block-diagonal SpMM on line-padded rows: tiles = lines (default) against the even split (SHADOW_SPMM_LINES=0)
export PYTHONPATH=.
for l in 0 1; do echo "LINES=$l"; SHADOW_SPMM_LINES=$l PAD=1 python scripts/probe_spmm.py 100 48 200; done
