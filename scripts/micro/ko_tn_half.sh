#!/bin/bash
# Knock-out: the interleaved weight-gradient step with three of its six products per tile (what two fp16 pieces per element
# would leave of the matrix-core work; results garbage)
cd "$(dirname "$0")/../.."
export PYTHONPATH=.
for v in "" "-DTN_KO_HALF" "" "-DTN_KO_HALF"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $v -Iinclude -Ishadow_gnn_amd/csrc shadow_gnn_amd/csrc/gemm.hip -o shadow_gnn_amd/csrc/_obj/gemm.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC shadow_gnn_amd/csrc/_obj/*.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  echo "variant [$v]: $(python scripts/probe_gemm_tn.py 2>&1 | grep 'M=289309 K=256 round 1')"
done
