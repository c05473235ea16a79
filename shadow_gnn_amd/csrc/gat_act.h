// gat_act.h -- the activation and head-slice reduction the GAT kernels (gat.hip) and the paired Linear's GAT tail
// (gemm_fused.hip, sl_gemm_nt2_gat_f32) must compute IDENTICALLY: the per-node attention terms u = att . act(z) come out of
// either place bit for bit.
#pragma once
#include "actnorm_common.h"

namespace shadow {

__device__ __forceinline__ float g_act_fwd(int act, float x) {
  switch (act) {
    case 1: return x > 0.f ? x : 0.f;
    case 2: return x > 0.f ? x : elu_neg(x);
    case 3: return tanhf(x);
    case 4: return x > 0.f ? x : 0.2f * x;
    default: return x;
  }
}
__device__ __forceinline__ float g_act_bwd(int act, float x, float h) {
  switch (act) {
    case 1: return x > 0.f ? 1.f : 0.f;
    case 2: return x > 0.f ? 1.f : h + 1.0f;
    case 3: return 1.f - h * h;
    case 4: return x > 0.f ? 1.f : 0.2f;
    default: return 1.f;
  }
}
// the same derivative from h = act(x) alone (x > 0 <=> h > 0 for relu / elu / leaky relu; at x = 0 both forms agree):
// the pre-activation need not be kept
__device__ __forceinline__ float g_act_bwd_h(int act, float h) {
  switch (act) {
    case 1: return h > 0.f ? 1.f : 0.f;
    case 2: return h > 0.f ? 1.f : h + 1.0f;
    case 3: return 1.f - h * h;
    case 4: return h > 0.f ? 1.f : 0.2f;
    default: return 1.f;
  }
}
__device__ __forceinline__ float4 act4(int act, float4 z) {
  return make_float4(g_act_fwd(act, z.x), g_act_fwd(act, z.y), g_act_fwd(act, z.z), g_act_fwd(act, z.w));
}
__device__ __forceinline__ float gat_dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// sum over the ls lanes of a head slice (ls power of two, wave-uniform): DPP butterflies (actnorm_common.h)
__device__ __forceinline__ float slice_sum(float v, uint32_t ls) {
  switch (ls) {
    case 1: return v;
    case 2: return group_sum<2>(v);
    case 4: return group_sum<4>(v);
    case 8: return group_sum<8>(v);
    case 16: return group_sum<16>(v);
    case 32: return group_sum<32>(v);
    default: return group_sum<64>(v);
  }
}

}  // namespace shadow
