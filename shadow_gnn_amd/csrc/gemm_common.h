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

// ---------------------------------------------------------------------------------------------------------------------
// Two-piece fp16 split (the nt kernels).  A row of an operand is first scaled by a power of two that puts its largest
// magnitude into [2^14, 2^15) -- the top of the fp16 range -- and every element x becomes h + m with h = rn16(x),
// m = rn16(x - h): 11 + 1 (sign of m) + 11 bits, i.e. |x - (h + m)| <= 2^-23 |x| for every element within 2^-17 of the
// row's maximum (both pieces normal), and <= 2^-39 of the row's maximum below that.  The product needs the three terms
// hh, hm, mh (mm <= 2^-22 |ab|, zero-mean under round-to-nearest): half the matrix-core work of the six-term bf16 form at
// the same measured error against fp64 (tests/test_layers_gpu.py::test_split_gemm_matches_fp64; scripts/probe_f16_split.py).
// The scale of a row is exact (power of two) and is taken out again when the accumulators leave the registers.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// power-of-two scale of a row whose largest magnitude is amax: amax * scale in [2^14, 2^15); exponent clamped to +-62 so
// that the product of two inverse scales stays finite (rows beyond 2^76 overflow to inf as they would soon after in fp32)
__device__ __forceinline__ float row_scale_of(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu) - 127;       // floor(log2(amax)) for normal values
  int sh = 14 - e;
  sh = amax > 0.f ? min(max(sh, -62), 62) : 0;
  return __uint_as_float((uint32_t)(127 + sh) << 23);
}

__device__ __forceinline__ void split8_f16(const float (&x)[8], float scale, half8 &h, half8 &m) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const float xs = x[j] * scale;
    const _Float16 hj = (_Float16)xs;                 // round to nearest even (v_cvt_pk_f16_f32)
    h[j] = hj;
    m[j] = (_Float16)(xs - (float)hj);
  }
}

// B-fragment reads of the nt kernels by hand: hipcc waits lgkmcnt(0) before every tile's MFMA group -- i.e. also for the
// next tile's reads it has just issued (~100 exposed cycles per tile) -- instead of a counted wait.  These reads are
// invisible to its scoreboard; the kernels wait with lds_wait<N>() (N = reads allowed to stay in flight; LDS returns in
// order) followed by a sched_barrier so that no MFMA is hoisted above the wait (cdna_hip_programming.md rule 18).
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
template <int kOffset, typename V = bf16x8>
__device__ __forceinline__ V lds_read_frag(uint32_t addr) {
  V v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(kOffset));
  return v;
}
template <int kOffset, typename V = bf16x8>
__device__ __forceinline__ void lds_write_frag(uint32_t addr, const V &v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(kOffset) : "memory");
}
// two dwords 256 * k0 and 256 * k1 bytes above addr (a column of a row tile with a 1 KB row pitch: k = 4 * row)
typedef float float2v __attribute__((ext_vector_type(2)));
template <int k0, int k1>
__device__ __forceinline__ float2v lds_read2st64(uint32_t addr) {
  float2v v;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(k0), "n"(k1));
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace shadow
