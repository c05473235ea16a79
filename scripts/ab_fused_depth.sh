#!/bin/bash
# A/B of the epilogue prefetch depths of csrc/gemm_fused.hip (first half / second half of the tile, backward / forward);
# rebuilds the one object file per variant on the GPU box
cd "$(dirname "$0")/.."
export PYTHONPATH=.
for v in "2 1 2 1" "2 1 4 1" "2 1 4 2" "2 1 8 2" "2 1 2 2" "2 1 2 4" "1 1 4 2"; do
  set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DSHADOW_EPI_DEPTH_BWD=$1 -DSHADOW_EPI_DEPTH_FWD=$2 -DSHADOW_EPI_DEPTH2_BWD=$3 -DSHADOW_EPI_DEPTH2_FWD=$4 \
      -Iinclude -Ishadow_gnn_amd/csrc shadow_gnn_amd/csrc/gemm_fused.hip -o shadow_gnn_amd/csrc/_obj/gemm_fused.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC shadow_gnn_amd/csrc/_obj/*.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  echo "depth bwd=$1/$3 fwd=$2/$4: $(python scripts/probe_fused_pair.py 2>&1 | tail -1)"
done
