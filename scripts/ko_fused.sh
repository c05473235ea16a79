#!/bin/bash
# Knock-out decomposition of the GEMM-epilogue kernels: rebuild gemm_fused.o with one part removed (results are garbage,
# only the time is of interest) and time both launches.  Runs on the GPU box.
cd "$(dirname "$0")/.."
export PYTHONPATH=.
for v in "" "-DFUSED_KO_MFMA" "-DFUSED_KO_SPLIT" "-DFUSED_KO_EPI" "-DFUSED_KO_BARRIER" "-DFUSED_KO_FILL" "-DFUSED_KO_ALOAD" "-DFUSED_KO_MFMA -DFUSED_KO_SPLIT" "-DFUSED_KO_MFMA -DFUSED_KO_EPI" "-DFUSED_KO_MFMA -DFUSED_KO_SPLIT -DFUSED_KO_EPI" "-DFUSED_KO_MFMA -DFUSED_KO_SPLIT -DFUSED_KO_EPI -DFUSED_KO_FILL" "-DFUSED_KO_MFMA -DFUSED_KO_SPLIT -DFUSED_KO_EPI -DFUSED_KO_FILL -DFUSED_KO_ALOAD"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $v -Iinclude -Ishadow_gnn_amd/csrc \
      shadow_gnn_amd/csrc/gemm_fused.hip -o shadow_gnn_amd/csrc/_obj/gemm_fused.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC shadow_gnn_amd/csrc/_obj/*.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  echo "variant [$v]: $(python scripts/ko_fused.py 2>&1 | tail -1)"
done
