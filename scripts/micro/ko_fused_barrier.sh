#!/bin/bash
# knock-out: the GEMM-epilogue kernels without their per-half-step workgroup barrier (racy, results garbage; time only)
cd "$(dirname "$0")/../.."
export PYTHONPATH=.
for v in "" "-DFUSED_KO_BARRIER" "-DFUSED_KO_BARRIER -DFUSED_KO_EPI" "-DFUSED_KO_EPI"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $v -Iinclude -Ishadow_gnn_amd/csrc shadow_gnn_amd/csrc/gemm_fused.hip -o shadow_gnn_amd/csrc/_obj/gemm_fused.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC shadow_gnn_amd/csrc/_obj/*.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  echo "variant [$v]: $(python scripts/ko_fused.py 2>&1 | tail -1)"
done
