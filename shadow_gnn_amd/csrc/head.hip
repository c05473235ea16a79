// head.hip -- the head of a node-classification step as three kernels instead of ~35.
//
// After the read-out (layers.py:159-163: the roots' rows of the last layer) DeepGNN.forward normalises every root embedding to
// unit length (shaDow/models.py:200), applies the one-layer classifier MLP(dim_hid -> num_classes, act 'I', norm_feat)
// (models.py:139-146, layers.py:376-400 with _f_norm_feat, layers.py:329-338) and DeepGNN.step takes the mean cross entropy over
// the roots (models.py:163-166).  On r = 16 .. 1024 rows that is a chain of ~15 forward and ~20 backward kernels of 3-10 us each
// on an idle chip, and as many host launches -- a tenth of a small-batch step.  Here:
//   head_fwd_kernel      one wavefront per root: L2 normalisation, the [C, F] product (W from LDS), norm over the classes,
//                        softmax, the root's loss (+ head_loss_kernel: their mean, added in a fixed order)
//   head_bwd_rows_kernel one wavefront per root: d preds -> norm backward -> dz, d emb (through the Linear and the normalisation)
//   head_bwd_cols_kernel dW = dz^T xn, dbias, dscale, doffset: a workgroup per (class, 256-row slice), the slices added in
//                        slice order by head_cols_finish_kernel
// Plain fp32 FMA arithmetic (2 r C F flop is nothing); lane layouts: FEATURE lanes hold the float4 at column 4 j of a row
// (F <= 256), CLASS lanes hold classes j, j + 64, j + 128, j + 192 (C <= 256).
#include "actnorm_common.h"
#include "common.h"

using namespace shadow;

namespace {

constexpr float kNormEps = 1e-12f;     // F.normalize(p = 2, dim = 1, eps = 1e-12)
constexpr float kVarEps = 1e-9f;       // _f_norm_feat, layers.py:335

struct HeadParams {
  const float *emb; int64_t lde;
  const float *W; int64_t ldw;
  const float *b, *scale, *offset;
  const int64_t *label;
  uint32_t r, F, C;
  float *xn, *z, *preds, *prob, *nrm, *rowloss, *loss;
  // backward
  const float *gloss;
  float *demb, *dz, *dp, *dph;
  float *dW, *db, *dscale, *doffset, *partial;
  uint32_t slices;
};

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// stage W [C, F] (pitch ldw) as dense [C, F] floats in LDS (F % 4 == 0, 16-byte aligned rows)
__device__ __forceinline__ void stage_w(const HeadParams &p, float *lds) {
  const uint32_t f4 = p.F >> 2;
  for (uint32_t i = threadIdx.x; i < p.C * f4; i += blockDim.x) {
    const uint32_t c = i / f4, k = (i - c * f4) << 2;
    st4(lds + (size_t)c * p.F + k, ld4(p.W + (size_t)c * p.ldw + k));
  }
  __syncthreads();
}

// mean and 1 / std over the C classes of the values the class lanes hold (zq[q] = class lane + 64 q)
__device__ __forceinline__ void class_stats(const float (&zq)[4], uint32_t lane, uint32_t C, float &mean, float &rstd) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) s += (lane + 64u * q < C) ? zq[q] : 0.f;
  mean = group_sum<64>(s) / (float)C;
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float d = zq[q] - mean;
    v += (lane + 64u * q < C) ? d * d : 0.f;
  }
  rstd = rsqrtf(group_sum<64>(v) / (float)C + kVarEps);
}

template <bool kLds>
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadParams p) {
  extern __shared__ float lds[];
  if (kLds) stage_w(p, lds);
  const float *Wp = kLds ? lds : p.W;
  const int64_t ldw = kLds ? (int64_t)p.F : p.ldw;
  const uint32_t lane = lane_id(), k4 = lane << 2;
  const bool fl = k4 < p.F;
  for (uint32_t i = blockIdx.x * 4 + wave_id(); i < p.r; i += gridDim.x * 4) {
    const float4 e = fl ? ld4(p.emb + (size_t)i * p.lde + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float raw = sqrtf(group_sum<64>(dot4(e, e)));
    const float nr = fmaxf(raw, kNormEps);
    const float4 x = make_float4(e.x / nr, e.y / nr, e.z / nr, e.w / nr);
    if (fl) st4(p.xn + (size_t)i * p.F + k4, x);
    float zq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      for (uint32_t cc = 0; cc < 64; ++cc) {
        const uint32_t c = 64u * q + cc;
        if (c >= p.C) break;
        const float part = fl ? dot4(x, ld4(Wp + (size_t)c * ldw + k4)) : 0.f;
        const float s = group_sum<64>(part) + (p.b ? p.b[c] : 0.f);
        if (lane == cc) zq[q] = s;
      }
    }
    float mean, rstd;
    class_stats(zq, lane, p.C, mean, rstd);
    float pq[4], mx = -INFINITY;
    const int64_t y = p.label[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = lane + 64u * q;
      const bool ok = c < p.C;
      pq[q] = ok ? (zq[q] - mean) * p.scale[c] * rstd + p.offset[c] : 0.f;          // (the reference's order of the products)
      if (ok) mx = fmaxf(mx, pq[q]);
    }
    mx = wave_max_f(mx);
    float se = 0.f, py = 0.f, ex[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = lane + 64u * q;
      ex[q] = c < p.C ? expf(pq[q] - mx) : 0.f;
      se += ex[q];
      py += (c < p.C && (int64_t)c == y) ? pq[q] : 0.f;
    }
    se = group_sum<64>(se);
    py = group_sum<64>(py);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = lane + 64u * q;
      if (c < p.C) {
        const size_t o = (size_t)i * p.C + c;
        p.z[o] = zq[q];
        p.preds[o] = pq[q];
        p.prob[o] = ex[q] / se;
      }
    }
    if (lane == 0) {
      p.nrm[i] = raw;
      p.rowloss[i] = mx + logf(se) - py;
    }
  }
}

// loss = mean of the r per-root losses, added in a fixed order by one workgroup.  (A separate launch on purpose: the single-kernel
// form -- the last workgroup to finish adds them, behind a device-scope fence per workgroup -- cost 41 us instead of 12 + 3 at
// 1 024 roots: on this chip such a fence writes the workgroup's whole L2 slice back, once per workgroup.)
__global__ __launch_bounds__(256) void head_loss_kernel(const float *__restrict__ rowloss, uint32_t r, float *__restrict__ loss) {
  __shared__ float red[256];
  float s = 0.f;
  for (uint32_t i = threadIdx.x; i < r; i += 256) s += rowloss[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / (float)r;
}

template <bool kLds>
__global__ __launch_bounds__(256) void head_bwd_rows_kernel(HeadParams p) {
  extern __shared__ float lds[];
  if (kLds) stage_w(p, lds);
  const float *Wp = kLds ? lds : p.W;
  const int64_t ldw = kLds ? (int64_t)p.F : p.ldw;
  const uint32_t lane = lane_id(), k4 = lane << 2;
  const bool fl = k4 < p.F;
  const float g = p.gloss[0] / (float)p.r;
  for (uint32_t i = blockIdx.x * 4 + wave_id(); i < p.r; i += gridDim.x * 4) {
    float zq[4], dzq[4];
    const int64_t y = p.label[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = lane + 64u * q;
      zq[q] = c < p.C ? p.z[(size_t)i * p.C + c] : 0.f;
    }
    float mean, rstd;
    class_stats(zq, lane, p.C, mean, rstd);
    float hh[4], dh[4], dpq[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = lane + 64u * q;
      const bool ok = c < p.C;
      hh[q] = ok ? (zq[q] - mean) * rstd : 0.f;
      dpq[q] = ok ? g * (p.prob[(size_t)i * p.C + c] - ((int64_t)c == y ? 1.f : 0.f)) : 0.f;
      dh[q] = ok ? dpq[q] * p.scale[c] : 0.f;
      s1 += dh[q];
      s2 += dh[q] * hh[q];
    }
    const float m1 = group_sum<64>(s1) / (float)p.C, m2 = group_sum<64>(s2) / (float)p.C;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = lane + 64u * q;
      dzq[q] = c < p.C ? rstd * (dh[q] - m1 - hh[q] * m2) : 0.f;
      if (c < p.C) {
        const size_t o = (size_t)i * p.C + c;
        p.dz[o] = dzq[q];
        p.dp[o] = dpq[q];
        p.dph[o] = dpq[q] * hh[q];
      }
    }
    // d xn = dz . W (feature lanes), then back through x / max(|x|, eps)
    float4 dx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      for (uint32_t cc = 0; cc < 64; ++cc) {
        const uint32_t c = 64u * q + cc;
        if (c >= p.C) break;
        const float d = __shfl(dzq[q], (int)cc, 64);
        if (fl) {
          const float4 w = ld4(Wp + (size_t)c * ldw + k4);
          dx.x += d * w.x; dx.y += d * w.y; dx.z += d * w.z; dx.w += d * w.w;
        }
      }
    }
    const float4 x = fl ? ld4(p.xn + (size_t)i * p.F + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float raw = p.nrm[i];
    const float dot = group_sum<64>(dot4(x, dx));
    float4 de;
    if (raw > kNormEps) {
      de = make_float4((dx.x - x.x * dot) / raw, (dx.y - x.y * dot) / raw, (dx.z - x.z * dot) / raw, (dx.w - x.w * dot) / raw);
    } else {
      de = make_float4(dx.x / kNormEps, dx.y / kNormEps, dx.z / kNormEps, dx.w / kNormEps);      // (the clamp passes no gradient)
    }
    if (fl) st4(p.demb + (size_t)i * p.F + k4, de);
  }
}

// One workgroup per slice of kColsRows rows, every class: thread t owns feature t.  The slice's dz / dp.h / dp values are staged
// in LDS class-major (each class's rows one after the other: the inner products read them as broadcast float4s), the thread's
// x values of the slice's rows sit in registers (independent loads); dW[c, t] of the slice = sum over its rows in row order,
// the slices are added in slice order by head_cols_finish_kernel.  (Until round 5: a workgroup per (class, 256-row slice)
// whose threads walked the 256 rows with one dependent load each -- 20 us alone, 58 us beside the sampler's kernels.)
constexpr uint32_t kColsRows = 32;
__global__ __launch_bounds__(256) void head_bwd_cols_kernel(HeadParams p) {
  extern __shared__ float lds[];                       // [3][C][kColsRows]
  const uint32_t sl = blockIdx.x, t = threadIdx.x, C = p.C;
  const uint32_t i0 = sl * kColsRows, rows = min(kColsRows, p.r - i0);
  float *ldz = lds, *ldph = lds + (size_t)C * kColsRows, *ldp = lds + (size_t)2 * C * kColsRows;
  for (uint32_t k = t; k < C * kColsRows; k += 256) {
    const uint32_t i = k / C, c = k - i * C;           // (consecutive threads read consecutive floats of the three arrays)
    const bool ok = i < rows;
    const size_t o = (size_t)(i0 + i) * C + c;
    ldz[c * kColsRows + i] = ok ? p.dz[o] : 0.f;
    ldph[c * kColsRows + i] = ok ? p.dph[o] : 0.f;
    ldp[c * kColsRows + i] = ok ? p.dp[o] : 0.f;
  }
  float x[kColsRows];
#pragma unroll
  for (uint32_t i = 0; i < kColsRows; ++i) x[i] = (t < p.F && i < rows) ? p.xn[(size_t)(i0 + i) * p.F + t] : 0.f;
  __syncthreads();
  // partial [slices][C][F + 4]; head_cols_finish_kernel adds the slices in slice order
  const uint32_t pw = p.F + 4;
  const bool direct = p.slices == 1;
  float *mine = p.partial + (size_t)sl * C * pw;
  for (uint32_t c = 0; c < C; ++c) {
    const float4 *dzc = reinterpret_cast<const float4 *>(ldz + c * kColsRows);
    float acc = 0.f;
#pragma unroll
    for (uint32_t i = 0; i < kColsRows / 4; ++i) {
      const float4 d = dzc[i];
      acc += d.x * x[4 * i]; acc += d.y * x[4 * i + 1]; acc += d.z * x[4 * i + 2]; acc += d.w * x[4 * i + 3];
    }
    if (t < p.F) {
      if (direct) p.dW[(size_t)c * p.F + t] = acc;
      else mine[(size_t)c * pw + t] = acc;
    }
  }
  for (uint32_t k = t; k < 3 * C; k += 256) {           // the three column sums of a class, rows in order
    const uint32_t which = k / C, c = k - which * C;
    const float *a = lds + ((size_t)which * C + c) * kColsRows;
    float s_ = 0.f;
#pragma unroll
    for (uint32_t i = 0; i < kColsRows; ++i) s_ += a[i];
    if (direct) (which == 0 ? p.db : which == 1 ? p.dscale : p.doffset)[c] = s_;
    else mine[(size_t)c * pw + p.F + which] = s_;
  }
}

__global__ __launch_bounds__(256) void head_cols_finish_kernel(HeadParams p) {
  const uint32_t c = blockIdx.x, pw = p.F + 4;
  const size_t qs = (size_t)p.C * pw;
  for (uint32_t k = threadIdx.x; k < pw - 1; k += 256) {
    const float *src = p.partial + (size_t)c * pw + k;
    float s = 0.f;
    uint32_t q = 0;
    for (; q + 8 <= p.slices; q += 8) {                     // (eight loads in flight, added in slice order)
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(q + j) * qs];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; q < p.slices; ++q) s += src[q * qs];
    if (k < p.F) p.dW[(size_t)c * p.F + k] = s;
    else if (k == p.F) p.db[c] = s;
    else if (k == p.F + 1) p.dscale[c] = s;
    else p.doffset[c] = s;
  }
}

int head_check(const char *who, uint32_t r, uint32_t F, uint32_t C, const void *emb, int64_t lde, const void *W, int64_t ldw) {
  if (r == 0 || F == 0 || C == 0 || F > 256 || (F & 3) || C > 256) return set_error(SG_ERR_INVALID, "%s: r = %u, F = %u (multiple of 4, <= 256), C = %u (<= 256)", who, r, F, C);
  if ((reinterpret_cast<uintptr_t>(emb) & 15) || (lde & 3) || (reinterpret_cast<uintptr_t>(W) & 15) || (ldw & 3))
    return set_error(SG_ERR_INVALID, "%s: rows must be 16-byte aligned", who);
  return SG_OK;
}

constexpr size_t kHeadLdsMax = 128 * 1024;

}  // namespace

extern "C" size_t sl_head_partial_floats(uint32_t r, uint32_t F, uint32_t C) {
  const size_t slices = (r + kColsRows - 1) / kColsRows;
  return slices > 1 ? slices * C * (size_t)(F + 4) : 1;
}

extern "C" int sl_head_fwd(const float *d_emb, int64_t lde, const float *d_W, int64_t ldw, const float *d_b, const float *d_scale,
                           const float *d_offset, const int64_t *d_label, uint32_t r, uint32_t F, uint32_t C, float *d_xn, float *d_z,
                           float *d_preds, float *d_prob, float *d_nrm, float *d_rowloss, float *d_loss, void *stream) {
  if (!d_emb || !d_W || !d_scale || !d_offset || !d_label || !d_xn || !d_z || !d_preds || !d_prob || !d_nrm || !d_rowloss || !d_loss)
    return set_error(SG_ERR_INVALID, "sl_head_fwd: null argument");
  int rc;
  if ((rc = head_check("sl_head_fwd", r, F, C, d_emb, lde, d_W, ldw)) != SG_OK) return rc;
  HeadParams p{};
  p.emb = d_emb; p.lde = lde; p.W = d_W; p.ldw = ldw; p.b = d_b; p.scale = d_scale; p.offset = d_offset; p.label = d_label;
  p.r = r; p.F = F; p.C = C; p.xn = d_xn; p.z = d_z; p.preds = d_preds; p.prob = d_prob; p.nrm = d_nrm; p.rowloss = d_rowloss;
  p.loss = d_loss;
  const size_t lds = (size_t)C * F * 4;
  const uint32_t grid = std::min<uint32_t>((r + 3) / 4, 512);
  SHD_PROF_FMT(4.0 * r * (2.0 * F + 3.0 * C) + 4.0 * C * F, 2.0 * r * C * F, stream, "head_fwd_F%u_C%u", F, C);
  if (lds <= kHeadLdsMax) {
    SHD_HIP(ensure_dynamic_lds((const void *)head_fwd_kernel<true>, lds));
    hipLaunchKernelGGL(head_fwd_kernel<true>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(head_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  }
  hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_rowloss, r, d_loss);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_head_bwd(const float *d_gloss, const float *d_xn, const float *d_z, const float *d_prob, const float *d_nrm,
                           const int64_t *d_label, const float *d_W, int64_t ldw, const float *d_scale, uint32_t r, uint32_t F,
                           uint32_t C, float *d_demb, float *d_dW, float *d_db, float *d_dscale, float *d_doffset, float *d_work,
                           float *d_partial, void *stream) {
  if (!d_gloss || !d_xn || !d_z || !d_prob || !d_nrm || !d_label || !d_W || !d_scale || !d_demb || !d_dW || !d_db || !d_dscale || !d_doffset ||
      !d_work || !d_partial)
    return set_error(SG_ERR_INVALID, "sl_head_bwd: null argument");
  int rc;
  if ((rc = head_check("sl_head_bwd", r, F, C, d_xn, F, d_W, ldw)) != SG_OK) return rc;
  HeadParams p{};
  p.W = d_W; p.ldw = ldw; p.scale = d_scale; p.label = d_label; p.r = r; p.F = F; p.C = C;
  p.xn = const_cast<float *>(d_xn); p.z = const_cast<float *>(d_z); p.prob = const_cast<float *>(d_prob); p.nrm = const_cast<float *>(d_nrm);
  p.gloss = d_gloss; p.demb = d_demb;
  p.dz = d_work; p.dp = d_work + (size_t)r * C; p.dph = d_work + 2 * (size_t)r * C;
  p.dW = d_dW; p.db = d_db; p.dscale = d_dscale; p.doffset = d_doffset; p.partial = d_partial;
  p.slices = (r + kColsRows - 1) / kColsRows;
  const size_t lds = (size_t)C * F * 4;
  const uint32_t grid = std::min<uint32_t>((r + 3) / 4, 512);
  {
    SHD_PROF_FMT(4.0 * r * (2.0 * F + 5.0 * C) + 4.0 * C * F, 2.0 * r * C * F, stream, "head_bwd_rows_F%u_C%u", F, C);
    if (lds <= kHeadLdsMax) {
      SHD_HIP(ensure_dynamic_lds((const void *)head_bwd_rows_kernel<true>, lds));
      hipLaunchKernelGGL(head_bwd_rows_kernel<true>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    } else {
      hipLaunchKernelGGL(head_bwd_rows_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    }
  }
  SHD_PROF_FMT(4.0 * r * (F + 3.0 * C) + 4.0 * C * F, 2.0 * r * C * F, stream, "head_bwd_cols_F%u_C%u", F, C);
  const size_t lds_cols = (size_t)3 * C * kColsRows * 4;
  SHD_HIP(ensure_dynamic_lds((const void *)head_bwd_cols_kernel, lds_cols));
  hipLaunchKernelGGL(head_bwd_cols_kernel, dim3(p.slices), dim3(256), lds_cols, (hipStream_t)stream, p);
  if (p.slices > 1) hipLaunchKernelGGL(head_cols_finish_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, p);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}
