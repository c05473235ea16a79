#!/bin/bash
# Per-phase shader-cycle stamps of one workgroup of gemm_nt_split_kernel (-DGEMM_TIMING build of csrc/gemm.hip):
# where a wavefront of the nt GEMM spends its time (prologue / operand split / MFMA tiles / counted waits / barrier / store).
cd "$(dirname "$0")/.."
cp shadow_gnn_amd/libshadow_hip.so /tmp/libshadow_hip.so.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DGEMM_TIMING $GEMM_EXP_FLAGS -Iinclude -Ishadow_gnn_amd/csrc shadow_gnn_amd/csrc/gemm.hip -o /tmp/gemm_timing.o || exit 1
objs=$(ls shadow_gnn_amd/csrc/_obj/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_timing.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
PYTHONPATH=. python - <<'PY'
import ctypes as C, torch
from shadow_gnn_amd import ops, _lib
lib = C.CDLL(_lib.LIB_PATH)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
for M, K in ((289309, 256), (289309, 512)):
    A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(256, K, device=dev, generator=g) / 16
    for _ in range(3): ops.mm_nt(A, W)
    buf = (C.c_ulonglong * 16)()
    lib.sl_gemm_debug_read(buf, 1)
    for _ in range(10): ops.mm_nt(A, W)
    lib.sl_gemm_debug_read(buf, 1)
    v = list(buf); n = max(1, v[8])
    names = ["prologue", "split", "mfma tiles", "counted wait", "barrier", "store"]
    tot = v[6] / n
    print(f"M={M} K={K}: workgroup lifetime {tot:.0f} shader cycles, {v[7]/n/100:.1f} us wall (100 MHz clock)")
    for i, nm in enumerate(names): print(f"   {nm:14s} {v[i]/n:9.0f} cycles  {100*v[i]/n/tot:5.1f} %")
PY
cp /tmp/libshadow_hip.so.keep shadow_gnn_amd/libshadow_hip.so
