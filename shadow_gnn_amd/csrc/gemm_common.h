// Shared pieces of the split-bf16 MFMA GEMMs (gemm.hip, gemm_fused.hip): fragment types and the exact
// three-way bf16 split of fp32 operands.
#pragma once
#include "common.h"

namespace shadow {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pack_hi16(uint32_t lo_src, uint32_t hi_src) {
  // [hi_src.b3, hi_src.b2, lo_src.b3, lo_src.b2]: two truncated-bf16 values in one dword
  return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u);
}

// exact 3-way split of 8 floats into bf16 pieces (truncation: every piece is a prefix of the
// remaining mantissa, so h + m + l == x bit for bit)
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
  uint32_t hb[8], mb[8], lb[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t xb = __float_as_uint(x[j]);
    hb[j] = xb & 0xFFFF0000u;
    const float r1 = x[j] - __uint_as_float(hb[j]);
    mb[j] = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mb[j]);
    lb[j] = __float_as_uint(r2);                 // <= 8 significant bits left: the top half holds them all
  }
  union { uint32_t u[4]; bf16x8 v; } H, Mm, L;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    H.u[j] = pack_hi16(hb[2 * j], hb[2 * j + 1]);
    Mm.u[j] = pack_hi16(mb[2 * j], mb[2 * j + 1]);
    L.u[j] = pack_hi16(lb[2 * j], lb[2 * j + 1]);
  }
  h = H.v; m = Mm.v; l = L.v;
}

// B-fragment reads of the nt kernels by hand: hipcc waits lgkmcnt(0) before every tile's MFMA group -- i.e. also for the
// next tile's reads it has just issued (~100 exposed cycles per tile) -- instead of a counted wait.  These reads are
// invisible to its scoreboard; the kernels wait with lds_wait<N>() (N = reads allowed to stay in flight; LDS returns in
// order) followed by a sched_barrier so that no MFMA is hoisted above the wait (cdna_hip_programming.md rule 18).
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
template <int kOffset>
__device__ __forceinline__ bf16x8 lds_read_frag(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(kOffset));
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace shadow
