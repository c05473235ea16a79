// Shared device helpers of the activation + feature-normalisation kernels (aggregate.hip) and of the GEMM kernels
// that carry the same arithmetic in their epilogues (gemm_fused.hip): vector accessors, the activation table of
// shaDow/layers.py:26-34 and the counter hash of the fused dropout.
#pragma once
#include "common.h"

namespace shadow {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// streaming variants (read once / written once, far larger than the caches): non-temporal hint.  Measured on the products
// benchmark (same box, A/B of two builds): act_norm forward 0.190 -> 0.187 ms, the rest unchanged; -DSHADOW_NO_NT_STREAM
// builds the plain accesses.
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4s(const float *p) {
#ifndef SHADOW_NO_NT_STREAM
  const v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return ld4(p);
#endif
}
__device__ __forceinline__ void st4s(float *p, float4 v) {
#ifndef SHADOW_NO_NT_STREAM
  v4f_nt w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<v4f_nt *>(p));
#else
  st4(p, v);
#endif
}

// e^x - 1 for the negative side of elu (round 6).  libm's expm1f is ~25 VALU instructions per element and elu sits in the
// epilogues / edge walks of kernels bound by VALU issue (the paired Linear's GAT tail: 512 of them per row); here the hardware
// exp2 where the result is not small (x <= -1/16: one ulp of e^x is below 1e-6 of e^x - 1) and four series terms near zero
// (next term x^5 / 120: 1.3e-7 relative at -1/16) -- about nine instructions, every kernel the same function (the bit-for-bit
// comparisons between alternative paths hold; against libm the difference is <= 1e-6 relative).  -DSHADOW_LIBM_ELU: expm1f.
__device__ __forceinline__ float elu_neg(float x) {
#ifdef SHADOW_LIBM_ELU
  return expm1f(x);
#else
  const float e = __builtin_amdgcn_exp2f(x * 1.44269504f) - 1.0f;
  const float s = x * (1.0f + x * (0.5f + x * (0.16666667f + x * 0.041666668f)));
  return x > -0.0625f ? s : e;
#endif
}

// F_ACT of shaDow/layers.py:26-34 (prelu variants carry parameters and stay in torch)
__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case 1: return x > 0.f ? x : 0.f;                       // relu
    case 2: return x > 0.f ? x : elu_neg(x);                // elu (alpha = 1)
    case 3: return tanhf(x);                                // tanh
    case 4: return x > 0.f ? x : 0.2f * x;                  // leakyrelu(0.2)
    default: return x;                                      // 0: identity ("I")
  }
}
// derivative given the input x and the output h = act(x)
__device__ __forceinline__ float act_bwd(int act, float x, float h) {
  switch (act) {
    case 1: return x > 0.f ? 1.f : 0.f;
    case 2: return x > 0.f ? 1.f : h + 1.0f;                // d/dx expm1(x) = exp(x) = h + 1
    case 3: return 1.f - h * h;
    case 4: return x > 0.f ? 1.f : 0.2f;
    default: return 1.f;
  }
}

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}


// All-reduce sums over aligned groups of LS = 2^k lanes on the DPP network: quad xor 1, quad xor 2, half-row mirror, row
// mirror (16 lanes), then one v_permlane16_swap (32) and one v_permlane32_swap (64) -- a handful of dependent VALU adds
// instead of k ds_bpermute round trips through the LDS crossbar (which made every row normalisation a latency chain).
#define SHD_DPP_ADD(v, CTRL) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false))
template <int LS>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(LS == 1 || LS == 2 || LS == 4 || LS == 8 || LS == 16 || LS == 32 || LS == 64, "power of two");
  if (LS >= 2) SHD_DPP_ADD(v, 0xB1);        // quad_perm [1, 0, 3, 2]
  if (LS >= 4) SHD_DPP_ADD(v, 0x4E);        // quad_perm [2, 3, 0, 1]
  if (LS >= 8) SHD_DPP_ADD(v, 0x141);       // row_half_mirror
  if (LS >= 16) SHD_DPP_ADD(v, 0x140);      // row_mirror
  if (LS >= 32) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  if (LS >= 64) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  return v;
}

// the same butterflies for a maximum over the whole wavefront
#define SHD_DPP_MAX(v, CTRL) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)))
__device__ __forceinline__ float wave_max_f(float v) {
  SHD_DPP_MAX(v, 0xB1); SHD_DPP_MAX(v, 0x4E); SHD_DPP_MAX(v, 0x141); SHD_DPP_MAX(v, 0x140);
  { const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])); }
  { const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])); }
  return v;
}
// ... and over aligned groups of LS lanes (the row maxima behind the fp16 operand scales, gemm_common.h)
template <int LS>
__device__ __forceinline__ float group_max(float v) {
  static_assert(LS == 1 || LS == 2 || LS == 4 || LS == 8 || LS == 16 || LS == 32 || LS == 64, "power of two");
  if (LS >= 2) SHD_DPP_MAX(v, 0xB1);
  if (LS >= 4) SHD_DPP_MAX(v, 0x4E);
  if (LS >= 8) SHD_DPP_MAX(v, 0x141);
  if (LS >= 16) SHD_DPP_MAX(v, 0x140);
  if (LS >= 32) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  if (LS >= 64) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  return v;
}
__device__ __forceinline__ float amax4(const float4 &v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// The fused dropout's mask rule (round 6: one murmur finaliser per TWO elements -- its two 32-bit multiplies run at quarter rate
// and the epilogues that carry the mask are bound by instruction issue; rounds 1 - 5 spent one finaliser per element).  Element
// (r, c) is kept iff the 16-bit field (c odd: high, c even: low) of
//   mix32(rowhash(r) + (c >> 1) * 0x9E3779B1),   rowhash(r) = mix32(r_lo ^ seed_lo) + r_hi + seed_hi          (32-bit wrap-around)
// is >= thr16 = clamp(floor(p * 65536), 1, 65535); kept values are scaled by 1 / (1 - p).  (The keep probability is
// 1 - thr16 / 65536: within 1.6e-5 of 1 - p.)  Restated in torch by ops.dropout_keep_mask.
__device__ __forceinline__ uint32_t drop_row_hash(uint32_t seed_lo, uint32_t seed_hi, uint64_t r) {
  return mix32((uint32_t)r ^ seed_lo) + (uint32_t)(r >> 32) + seed_hi;
}
// keep-mask (bit k: component k) of the float4 at column f (f % 4 == 0) of a row with hash `rowh`
__device__ __forceinline__ uint32_t drop_keep4_h(uint32_t rowh, uint32_t thr16, uint32_t f) {
  const uint32_t base = rowh + (f >> 1) * 0x9E3779B1u;
  const uint32_t h0 = mix32(base), h1 = mix32(base + 0x9E3779B1u);
  return ((h0 & 0xFFFFu) >= thr16 ? 1u : 0u) | ((h0 >> 16) >= thr16 ? 2u : 0u) | ((h1 & 0xFFFFu) >= thr16 ? 4u : 0u) |
         ((h1 >> 16) >= thr16 ? 8u : 0u);
}
__device__ __forceinline__ uint32_t drop_keep4_raw(uint32_t seed_lo, uint32_t seed_hi, uint32_t thr16, uint64_t r, uint32_t f) {
  return drop_keep4_h(drop_row_hash(seed_lo, seed_hi, r), thr16, f);
}
// host side: the threshold of a dropout probability 0 < p < 1
inline uint32_t drop_threshold16(float drop_p) {
  const double t = (double)drop_p * 65536.0;
  return (uint32_t)(t < 1.0 ? 1.0 : (t > 65535.0 ? 65535.0 : t));
}

}  // namespace shadow
