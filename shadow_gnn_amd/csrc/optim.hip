// optim.hip -- clip-by-global-norm + Adam over one flat fp32 buffer in two launches.
//
// shaDow/models.py:225-226 ends every training step with torch.nn.utils.clip_grad_norm_(parameters, 5) and
// optimizer.step() (torch.optim.Adam, default betas / eps, no weight decay, no amsgrad).  On the flat gradient /
// parameter / moment buffers of optim.FlatAdam that is a dozen element-wise torch kernels over ~0.6 M floats -- pure
// launch latency, and at the reference's own batch sizes a visible part of the step.  Here: one pass for the squared
// norm (per-block partial sums, summed by every block of the second pass in block order: deterministic, no float
// atomics), one pass that scales the gradient in place and applies the Adam update, statement for statement what torch
// computes:
//     g  = g * min(1, max_norm / (||g|| + 1e-6))
//     m  = m + (g - m) * (1 - b1)                         (Tensor.lerp_, weight < 0.5)
//     v  = v * b2 + (1 - b2) * g * g
//     p  = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <math.h>

#include <algorithm>

#include "common.h"

using namespace shadow;

namespace {

constexpr int kOptBlock = 256;
constexpr int kOptMaxBlocks = 256;          // partial sums of the norm pass (and the most blocks it launches)

__global__ void __launch_bounds__(kOptBlock) sqnorm_partial_kernel(const float *__restrict__ g, uint64_t n, float *__restrict__ part) {
  __shared__ float red[kOptBlock / 64];
  float acc = 0.f;
  const uint64_t n4 = n >> 2;
  for (uint64_t i = (uint64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * kOptBlock) {
    const float4 x = reinterpret_cast<const float4 *>(g)[i];
    acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float x = g[(n4 << 2) + threadIdx.x]; acc += x * x; }
  acc = wave_reduce_sum_f(acc);
  if (lane_id() == 0) red[wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kOptBlock / 64; w++) s += red[w];
    part[blockIdx.x] = s;
  }
}

struct AdamArgs {
  float *p, *g, *m, *v;
  uint64_t n;
  float step_size, one_m_b1, b2, one_m_b2, inv_sqrt_bc2, eps, max_norm;
  const float *part;
  uint32_t nparts;
  float *out_norm;
};

__device__ __forceinline__ void adam1(float &p, float &g, float &m, float &v, float coef, const AdamArgs &a) {
  g *= coef;
  m = m + (g - m) * a.one_m_b1;
  v = v * a.b2 + a.one_m_b2 * g * g;
  p = p - a.step_size * (m / (sqrtf(v) * a.inv_sqrt_bc2 + a.eps));
}

__global__ void __launch_bounds__(kOptBlock) clip_adam_kernel(AdamArgs a) {
  float coef = 1.f;
  if (a.part) {
    // every block adds the partial sums in the same order: same total everywhere, no second launch
    float tot = 0.f;
    for (uint32_t i = 0; i < a.nparts; i++) tot += a.part[i];
    const float norm = sqrtf(tot);
    if (a.max_norm > 0.f) coef = fminf(1.f, a.max_norm / (norm + 1e-6f));
    if (a.out_norm && blockIdx.x == 0 && threadIdx.x == 0) *a.out_norm = norm;
  }
  const uint64_t n4 = a.n >> 2;
  for (uint64_t i = (uint64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * kOptBlock) {
    float4 p = reinterpret_cast<float4 *>(a.p)[i], g = reinterpret_cast<float4 *>(a.g)[i];
    float4 m = reinterpret_cast<float4 *>(a.m)[i], v = reinterpret_cast<float4 *>(a.v)[i];
    adam1(p.x, g.x, m.x, v.x, coef, a); adam1(p.y, g.y, m.y, v.y, coef, a);
    adam1(p.z, g.z, m.z, v.z, coef, a); adam1(p.w, g.w, m.w, v.w, coef, a);
    reinterpret_cast<float4 *>(a.p)[i] = p; reinterpret_cast<float4 *>(a.g)[i] = g;
    reinterpret_cast<float4 *>(a.m)[i] = m; reinterpret_cast<float4 *>(a.v)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const uint64_t i = (n4 << 2) + threadIdx.x;
    adam1(a.p[i], a.g[i], a.m[i], a.v[i], coef, a);
  }
}

}  // namespace

extern "C" uint32_t sl_clip_adam_scratch_floats(void) { return kOptMaxBlocks + 1; }

extern "C" int sl_clip_adam(float *d_param, float *d_grad, float *d_exp_avg, float *d_exp_avg_sq, uint64_t n, double lr,
                            double beta1, double beta2, double eps, uint32_t step, float max_norm, float *d_scratch,
                            void *stream) {
  if (!d_param || !d_grad || !d_exp_avg || !d_exp_avg_sq || !d_scratch)
    return set_error(SG_ERR_INVALID, "sl_clip_adam: null argument");
  if (step == 0) return set_error(SG_ERR_INVALID, "sl_clip_adam: step counts from 1");
  if ((reinterpret_cast<uintptr_t>(d_param) | reinterpret_cast<uintptr_t>(d_grad) | reinterpret_cast<uintptr_t>(d_exp_avg) |
       reinterpret_cast<uintptr_t>(d_exp_avg_sq)) & 15)
    return set_error(SG_ERR_INVALID, "sl_clip_adam: buffers must be 16-byte aligned");
  if (n == 0) return SG_OK;
  hipStream_t st = (hipStream_t)stream;
  const uint64_t n4 = (n + 3) >> 2;
  const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(kOptMaxBlocks, (n4 + kOptBlock - 1) / kOptBlock));
  AdamArgs a;
  a.p = d_param; a.g = d_grad; a.m = d_exp_avg; a.v = d_exp_avg_sq; a.n = n;
  // double precision for the bias corrections, as Python does them
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.step_size = (float)(lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  // (1 - beta) in double precision, then rounded: what Python hands to lerp_ / addcmul_
  a.one_m_b1 = (float)(1.0 - beta1); a.one_m_b2 = (float)(1.0 - beta2);
  a.b2 = (float)beta2; a.eps = (float)eps; a.max_norm = max_norm;
  a.part = d_scratch; a.nparts = blocks; a.out_norm = d_scratch + kOptMaxBlocks;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(blocks), dim3(kOptBlock), 0, st, d_grad, n, d_scratch);
  SHD_HIP(hipGetLastError());
  hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(kOptBlock), 0, st, a);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}
