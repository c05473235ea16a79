#!/usr/bin/env bash
# A/B of the two scan kernels for plain calls in ONE gpurun call (boxes differ by several per cent): flat run list vs
# row windows, alternating, sample kernel time from scripts/probe_sampler_batch.py.
export PYTHONPATH="${GRAFT_REPO_ROOT:-$PWD}"
for rep in 1 2; do
  for impl in flat window; do
    echo "== $impl"; SHADOW_SG_SCAN_IMPL=$impl timeout 200 python scripts/probe_sampler_batch.py ${@:-1024 8192} 2>&1 | grep "B="
  done
done
