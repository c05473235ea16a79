#!/bin/bash
# Knock-out decomposition of the cooperative weight-gradient kernel: rebuild gemm.o with one part removed (results are
# garbage, only the time is of interest).  Runs on the GPU box.  Round 3, M = 289 k, N = K = 256: full 233 us; without the
# MFMAs 164; without the operand split 200; without the row-tile copies 194; without all three 51 (launch, LDS zeroing,
# partial store + reduce).
cd "$(dirname "$0")/.."
export PYTHONPATH=.
for v in "" "-DTN_FILL_SPREAD" "-DTN_KO_MFMA" "-DTN_KO_SPLIT" "-DTN_KO_FILL" "-DTN_KO_BARRIER" "-DTN_KO_MFMA -DTN_KO_SPLIT" "-DTN_KO_MFMA -DTN_KO_SPLIT -DTN_KO_FILL"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $v -Iinclude -Ishadow_gnn_amd/csrc \
      shadow_gnn_amd/csrc/gemm.hip -o shadow_gnn_amd/csrc/_obj/gemm.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC shadow_gnn_amd/csrc/_obj/*.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  echo "variant [$v]: $(python scripts/probe_gemm_tn.py 2>&1 | grep 'K=256 round 2' )"
done
