#!/bin/bash
# Premise of a role-split GEMM-epilogue kernel: does the main loop need two wavefronts per SIMD?  The two launches with and
# without their epilogue, at two workgroups per CU (as shipped) and at one (LDS request raised: FUSED_ONE_WG).  GPU box.
cd "$(dirname "$0")/../.."
export PYTHONPATH=.
for v in "" "-DFUSED_KO_EPI" "-DFUSED_ONE_WG" "-DFUSED_ONE_WG -DFUSED_KO_EPI" "-DFUSED_ONE_WG -DFUSED_KO_EPI -DFUSED_KO_MFMA" "-DFUSED_KO_EPI -DFUSED_KO_MFMA"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $v -Iinclude -Ishadow_gnn_amd/csrc \
      shadow_gnn_amd/csrc/gemm_fused.hip -o /tmp/gemm_fused_ko.o || exit 1
  objs=$(ls shadow_gnn_amd/csrc/_obj/*.o | grep -v gemm_fused.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_fused_ko.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  echo "variant [$v]: $(python scripts/ko_fused.py 2>&1 | tail -1)"
done
