// Readout and input-encoding kernels of the shaDow model around the conv stack (gfx950).
//
//   sl_segment_pool_fwd / _bwd   mean / max / sum over the rows of each subgraph
//                                (ResPool, shaDow/layers.py:166-183: F.embedding_bag over
//                                 the subgraph offsets)
//   sl_encode_codes              hop / ppr / drnl -> per-node bit mask of the active one-hot
//                                columns (frontend/graph.py:134-172)
//   sl_onehot_linear_fwd / _bwd  out = X + onehot @ W^T + b without materialising the one-hot
//                                matrix (the feature-augmentation Linear, shaDow/models.py:
//                                DeepGNN.forward "feat_aug_emb"; sum mode)
//
// All kernels are HBM-streaming: one pass over the [n, F] operand, float accumulation in
// registers, deterministic two-stage reductions (no float atomics).
#include <algorithm>

#include "common.h"

namespace shadow {
namespace {

constexpr uint32_t kPB = 256;    // threads per block (4 wavefronts)

__device__ __forceinline__ float ldf(const float *p, uint32_t f, uint32_t F) { return f < F ? p[f] : 0.f; }

// ---------------------------------------------------------------- segment pooling
// grid (P, ceil(F/256)); thread t owns feature f = chunk*256 + (t & 63) * 4 .. +3 of the rows
// a + wave, a + wave + 4, ...; the four wavefronts are combined through LDS.
// mode 0 mean, 1 max, 2 sum.
__global__ void __launch_bounds__(kPB)
segment_pool_fwd_kernel(const float *__restrict__ X, int64_t ldx, const uint32_t *__restrict__ node_off,
                        uint32_t F, int mode, float *__restrict__ out, int64_t ldo,
                        uint32_t *__restrict__ argmax) {
  __shared__ float4 red[4][64];
  __shared__ uint4 redi[4][64];
  const uint32_t s = blockIdx.x;
  const uint32_t a = node_off[s], ns = node_off[s + 1] - a;
  const uint32_t lane = lane_id(), wv = wave_id();
  const uint32_t f = blockIdx.y * 256u + lane * 4u;
  const bool vec = (f + 3 < F) && ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  const float ninf = -__builtin_huge_valf();
  float4 acc = (mode == 1) ? make_float4(ninf, ninf, ninf, ninf) : make_float4(0.f, 0.f, 0.f, 0.f);
  uint4 am = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  for (uint32_t i = wv; i < ns; i += 8) {
    // two rows in flight per wavefront
    const uint32_t i1 = i + 4;
    const float *r0 = X + (int64_t)(a + i) * ldx, *r1 = X + (int64_t)(a + min(i1, ns - 1)) * ldx;
    float4 v0, v1;
    if (vec) { v0 = *reinterpret_cast<const float4 *>(r0 + f); v1 = *reinterpret_cast<const float4 *>(r1 + f); }
    else {
      v0 = make_float4(ldf(r0, f, F), ldf(r0, f + 1, F), ldf(r0, f + 2, F), ldf(r0, f + 3, F));
      v1 = make_float4(ldf(r1, f, F), ldf(r1, f + 1, F), ldf(r1, f + 2, F), ldf(r1, f + 3, F));
    }
    if (mode == 1) {
      if (v0.x > acc.x) { acc.x = v0.x; am.x = a + i; }
      if (v0.y > acc.y) { acc.y = v0.y; am.y = a + i; }
      if (v0.z > acc.z) { acc.z = v0.z; am.z = a + i; }
      if (v0.w > acc.w) { acc.w = v0.w; am.w = a + i; }
      if (i1 < ns) {
        if (v1.x > acc.x) { acc.x = v1.x; am.x = a + i1; }
        if (v1.y > acc.y) { acc.y = v1.y; am.y = a + i1; }
        if (v1.z > acc.z) { acc.z = v1.z; am.z = a + i1; }
        if (v1.w > acc.w) { acc.w = v1.w; am.w = a + i1; }
      }
    } else {
      acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      if (i1 < ns) { acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w; }
    }
  }
  red[wv][lane] = acc;
  redi[wv][lane] = am;
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int w = 1; w < 4; w++) {
      const float4 o = red[w][lane];
      const uint4 oi = redi[w][lane];
      if (mode == 1) {
        // ties go to the smaller row id (the first maximum, as a sequential scan would pick)
        if (o.x > acc.x || (o.x == acc.x && oi.x < am.x)) { acc.x = o.x; am.x = oi.x; }
        if (o.y > acc.y || (o.y == acc.y && oi.y < am.y)) { acc.y = o.y; am.y = oi.y; }
        if (o.z > acc.z || (o.z == acc.z && oi.z < am.z)) { acc.z = o.z; am.z = oi.z; }
        if (o.w > acc.w || (o.w == acc.w && oi.w < am.w)) { acc.w = o.w; am.w = oi.w; }
      } else {
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
      }
    }
    if (ns == 0) acc = make_float4(0.f, 0.f, 0.f, 0.f);              // empty bag -> zeros
    if (mode == 0 && ns) { const float inv = 1.0f / (float)ns; acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv; }
    float *o = out + (int64_t)s * ldo;
    const float av[4] = {acc.x, acc.y, acc.z, acc.w};
    const uint32_t iv[4] = {am.x, am.y, am.z, am.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (f + k < F) {
        o[f + k] = av[k];
        if (argmax) argmax[(size_t)s * F + f + k] = iv[k];
      }
    }
  }
}

__global__ void __launch_bounds__(kPB)
segment_pool_bwd_kernel(const float *__restrict__ dout, int64_t lddo, const uint32_t *__restrict__ node_off,
                        uint32_t F, int mode, const uint32_t *__restrict__ argmax, float *__restrict__ dX,
                        int64_t lddx) {
  const uint32_t s = blockIdx.x;
  const uint32_t a = node_off[s], ns = node_off[s + 1] - a;
  const uint32_t lane = lane_id(), wv = wave_id();
  const uint32_t f = blockIdx.y * 256u + lane * 4u;
  if (ns == 0) return;
  float g[4];
  uint32_t am[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    g[k] = (f + k < F) ? dout[(int64_t)s * lddo + f + k] : 0.f;
    am[k] = (mode == 1 && f + k < F) ? argmax[(size_t)s * F + f + k] : 0u;
  }
  if (mode == 0) { const float inv = 1.0f / (float)ns;
#pragma unroll
    for (int k = 0; k < 4; k++) g[k] *= inv; }
  for (uint32_t i = wv; i < ns; i += 4) {
    float *r = dX + (int64_t)(a + i) * lddx;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (f + k < F) r[f + k] = (mode == 1) ? ((am[k] == a + i) ? g[k] : 0.f) : g[k];
  }
}

// The same gradient as a TABLE: row s < P = dout[s, :] (/ n_s for the mean) -- what every non-root row of subgraph s receives --
// and row P + k = that of root k's subgraph + droots[k, :] (+ the droots of every other root on the same row, in index order:
// index_add_'s sum), with idx[i] = the table row of batch row i.  A consumer that adds the gradient inside its own pass
// (sl_gemm_an_bwd_plain) reads the 2 MB table instead of an [n, F] expansion somebody had to write first.
__device__ __forceinline__ uint32_t seg_of_row(const uint32_t *__restrict__ node_off, uint32_t P, uint32_t r) {
  uint32_t lo = 0, hi = P;                      // largest s with node_off[s] <= r (empty subgraphs share an offset: the last wins)
  while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (node_off[mid] <= r) lo = mid; else hi = mid; }
  return lo;
}

__global__ void pool_row_index_kernel(const uint32_t *__restrict__ node_off, uint32_t P, uint32_t n, uint32_t *__restrict__ idx) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) idx[r] = seg_of_row(node_off, P, r);
}

__global__ void pool_root_index_kernel(const int64_t *__restrict__ rows, uint32_t K, uint32_t P, uint32_t n, uint32_t *__restrict__ idx) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K && (uint64_t)rows[k] < (uint64_t)n) atomicMax(&idx[rows[k]], P + k);        // (a row listed twice: its last entry's table row)
}

__global__ void __launch_bounds__(64)
pool_grad_table_kernel(const float *__restrict__ dout, int64_t lddo, const float *__restrict__ droots, int64_t lddr,
                       const uint32_t *__restrict__ node_off, uint32_t P, const int64_t *__restrict__ rows, uint32_t K, uint32_t n,
                       uint32_t F, int mode, float *__restrict__ table, int64_t ldt) {
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  uint32_t s = b;
  int64_t r = -1;
  if (b >= P) {
    r = rows[b - P];
    if ((uint64_t)r >= (uint64_t)n) return;
    s = seg_of_row(node_off, P, (uint32_t)r);
  }
  const uint32_t ns = node_off[s + 1] - node_off[s];
  const float inv = (mode == 0 && ns) ? 1.0f / (float)ns : 1.0f;
  for (uint32_t f = lane; f < F; f += 64) {
    float g = dout ? dout[(int64_t)s * lddo + f] : 0.f;
    if (mode == 0) g *= inv;
    table[(int64_t)b * ldt + f] = g;
  }
  if (b < P || !droots) return;
  // the roots on row r, in index order (the wavefront looks at 64 entries at a time)
  for (uint32_t k0 = 0; k0 < K; k0 += 64) {
    const uint32_t k = k0 + lane;
    unsigned long long m = __ballot(k < K && rows[k] == r);
    while (m) {
      const uint32_t kk = k0 + (uint32_t)__builtin_ctzll(m);
      m &= m - 1;
      for (uint32_t f = lane; f < F; f += 64) table[(int64_t)b * ldt + f] += droots[(int64_t)kk * lddr + f];
    }
  }
}

// ---------------------------------------------------------------- encodings
// kind 0: hops (uint32; 0xFFFFFFFF = unreachable)   graph.py:134-147
//      1: pprs (float)                              graph.py:149-159 (bins can overlap at their edges)
//      2: drnls (uint32)                            graph.py:161-172
__global__ void encode_codes_kernel(int kind, const void *__restrict__ src, uint32_t n, uint32_t dim,
                                    uint32_t *__restrict__ codes) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // every kind reads one 32-bit word per node
  const uint32_t raw = reinterpret_cast<const uint32_t *>(src)[i];
  uint32_t code = 0;
  if (kind == 0) {
    const int64_t hs = (raw == 0xFFFFFFFFu) ? -1 : (int64_t)raw;   // the reference holds hop as a signed integer
    if (hs >= -1 && hs <= (int64_t)dim - 2) code |= 1u << (uint32_t)(hs + 1);
    if (hs >= 255) code |= 1u;
  } else if (kind == 1) {
    const double p = (double)__uint_as_float(raw);
    // cond_filter[c] = 0.25^c (c < dim), cond_filter[dim] = 0
    double hi = 1.0;
    for (uint32_t c = 0; c < dim; c++) {
      const double lo = (c + 1 < dim) ? hi * 0.25 : 0.0;
      if (p <= hi && p >= lo) code |= 1u << c;
      hi = hi * 0.25;
    }
  } else {
    uint32_t v = raw;
    if (v >= 255u) v = 0;
    if (v > dim - 1) v = 0;
    code = 1u << v;
  }
  codes[i] = code;
}

constexpr int kMaxDim = 16;      // one-hot width handled by the fused Linear (reference: 7..)

// out[i,:] = (X ? X[i,:] : 0) + b + sum_{c in codes[i]} Wt[c,:]      Wt = W^T, [dim, F]
__global__ void __launch_bounds__(kPB)
onehot_linear_fwd_kernel(const float *__restrict__ X, int64_t ldx, const uint32_t *__restrict__ codes,
                         const float *__restrict__ Wt, const float *__restrict__ bias, uint32_t n, uint32_t F,
                         uint32_t dim, float *__restrict__ out, int64_t ldo) {
  const uint32_t per_row = (F + 3) / 4;                                 // threads per row
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t i = t / per_row;
  const uint32_t f = (uint32_t)(t % per_row) * 4;
  if (i >= n) return;
  const uint32_t code = codes[i];
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = (f + k < F) ? ((X ? X[(int64_t)i * ldx + f + k] : 0.f) + (bias ? bias[f + k] : 0.f)) : 0.f;
  for (uint32_t c = 0; c < dim; c++) {
    if ((code >> c) & 1u) {
#pragma unroll
      for (int k = 0; k < 4; k++) if (f + k < F) v[k] += Wt[(size_t)c * F + f + k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) if (f + k < F) out[(int64_t)i * ldo + f + k] = v[k];
}

// partial[blk][c][f] = sum over the block's rows with bit c set of dOut[i,f]; c == dim: all rows (bias)
__global__ void __launch_bounds__(kPB)
onehot_linear_bwd_kernel(const float *__restrict__ dout, int64_t lddo, const uint32_t *__restrict__ codes,
                         uint32_t n, uint32_t F, uint32_t dim, float *__restrict__ partial) {
  // thread = (row group, feature); 256 threads = (256 / FT) row groups x FT features
  const uint32_t FT = F >= 256 ? 256 : (F >= 128 ? 128 : 64);
  const uint32_t fl = threadIdx.x % FT, rgp = threadIdx.x / FT, nrg = kPB / FT;
  __shared__ float red[kPB];
  for (uint32_t f0 = 0; f0 < F; f0 += FT) {
    const uint32_t f = f0 + fl;
    float acc[kMaxDim + 1];
#pragma unroll
    for (int c = 0; c <= kMaxDim; c++) acc[c] = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * nrg + rgp; i < n; i += (uint64_t)gridDim.x * nrg) {
      const uint32_t code = codes[i];
      const float g = f < F ? dout[(int64_t)i * lddo + f] : 0.f;
#pragma unroll
      for (int c = 0; c < kMaxDim; c++) acc[c] += ((code >> c) & 1u) ? g : 0.f;
      acc[kMaxDim] += g;
    }
#pragma unroll
    for (int c = 0; c <= kMaxDim; c++) {
      if (c < (int)dim || c == kMaxDim) {
        __syncthreads();
        red[threadIdx.x] = acc[c];
        __syncthreads();
        if (rgp == 0 && f < F) {
          float sum = 0.f;
          for (uint32_t q = 0; q < nrg; q++) sum += red[q * FT + fl];
          const uint32_t cc = (c == kMaxDim) ? dim : (uint32_t)c;
          partial[((size_t)blockIdx.x * (dim + 1) + cc) * F + f] = sum;
        }
      }
    }
  }
}

// dWt[c,f] (c < dim) and db[f] (c == dim): ordered sum of the block partials
// dWt / db = sum over the blocks' partial rows, fixed order (deterministic).  A workgroup owns 32 outputs; 32 slices of
// the block range per output, four running sums per thread, slices combined through LDS in slice order (one thread per
// output over all blocks was a 50 us chain of L2 round trips).
constexpr int kOhCols = 32, kOhSlices = 32;
__global__ void __launch_bounds__(kOhCols * kOhSlices)
onehot_linear_finish_kernel(const float *__restrict__ partial, uint32_t nblocks, uint32_t F, uint32_t dim,
                            float *__restrict__ dWt, float *__restrict__ db) {
  __shared__ float red[kOhSlices][kOhCols];
  const uint32_t col = threadIdx.x % kOhCols, sl = threadIdx.x / kOhCols;
  const uint32_t t = blockIdx.x * kOhCols + col, total = (dim + 1) * F;
  const uint32_t per = (nblocks + kOhSlices - 1) / kOhSlices;
  const uint32_t b0 = sl * per, b1 = min(nblocks, b0 + per);
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
  if (t < total) {
    uint32_t b = b0;
    for (; b + 4 <= b1; b += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) a4[u] += partial[(size_t)(b + u) * total + t];
    }
    for (int u = 0; b < b1; b++, u++) a4[u] += partial[(size_t)b * total + t];
  }
  red[sl][col] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  __syncthreads();
  if (sl == 0 && t < total) {
    float sum = 0.f;
    for (int q = 0; q < kOhSlices; q++) sum += red[q][col];
    const uint32_t c = t / F, f = t % F;
    if (c < dim) dWt[(size_t)c * F + f] = sum;
    else if (db) db[f] = sum;
  }
}

}  // namespace
}  // namespace shadow

using namespace shadow;

extern "C" int sl_segment_pool_fwd(const float *d_X, int64_t ldx, const uint32_t *d_node_off, uint32_t num_subg,
                                   uint32_t F, int mode, float *d_out, int64_t ldo, uint32_t *d_argmax,
                                   void *stream) {
  if (!d_X || !d_node_off || !d_out) return set_error(SG_ERR_INVALID, "sl_segment_pool_fwd: null argument");
  if (mode < 0 || mode > 2) return set_error(SG_ERR_INVALID, "sl_segment_pool_fwd: mode %d (0 mean, 1 max, 2 sum)", mode);
  if (num_subg == 0 || F == 0) return SG_OK;
  hipLaunchKernelGGL(segment_pool_fwd_kernel, dim3(num_subg, (F + 255) / 256), dim3(kPB), 0, (hipStream_t)stream, d_X,
                     ldx, d_node_off, F, mode, d_out, ldo, mode == 1 ? d_argmax : nullptr);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_segment_pool_bwd(const float *d_dout, int64_t lddo, const uint32_t *d_node_off, uint32_t num_subg,
                                   uint32_t F, int mode, const uint32_t *d_argmax, float *d_dX, int64_t lddx,
                                   void *stream) {
  if (!d_dout || !d_node_off || !d_dX) return set_error(SG_ERR_INVALID, "sl_segment_pool_bwd: null argument");
  if (mode < 0 || mode > 2) return set_error(SG_ERR_INVALID, "sl_segment_pool_bwd: mode %d", mode);
  if (mode == 1 && !d_argmax) return set_error(SG_ERR_INVALID, "sl_segment_pool_bwd: max pooling needs argmax");
  if (num_subg == 0 || F == 0) return SG_OK;
  hipLaunchKernelGGL(segment_pool_bwd_kernel, dim3(num_subg, (F + 255) / 256), dim3(kPB), 0, (hipStream_t)stream,
                     d_dout, lddo, d_node_off, F, mode, d_argmax, d_dX, lddx);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_pool_grad_rows(const uint32_t *d_node_off, uint32_t num_subg, const int64_t *d_rows, uint32_t num_roots, uint32_t n,
                                 uint32_t *d_index, void *stream) {
  if (!d_node_off || !d_index || (num_roots && !d_rows)) return set_error(SG_ERR_INVALID, "sl_pool_grad_rows: null argument");
  if (num_subg == 0) return set_error(SG_ERR_INVALID, "sl_pool_grad_rows: no subgraph");
  if (n == 0) return SG_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pool_row_index_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_node_off, num_subg, n, d_index);
  SHD_HIP(hipGetLastError());
  if (num_roots) {
    hipLaunchKernelGGL(pool_root_index_kernel, dim3((num_roots + 255) / 256), dim3(256), 0, st, d_rows, num_roots, num_subg, n, d_index);
    SHD_HIP(hipGetLastError());
  }
  return SG_OK;
}

extern "C" int sl_pool_grad_table(const float *d_dout, int64_t lddo, const float *d_droots, int64_t lddr, const uint32_t *d_node_off,
                                  uint32_t num_subg, const int64_t *d_rows, uint32_t num_roots, uint32_t n, uint32_t F, int mode,
                                  float *d_table, int64_t ldt, void *stream) {
  if (!d_node_off || !d_table || (num_roots && !d_rows)) return set_error(SG_ERR_INVALID, "sl_pool_grad_table: null argument");
  if (mode != 0 && mode != 2) return set_error(SG_ERR_INVALID, "sl_pool_grad_table: mode %d (mean 0 / sum 2: the max's gradient is not one row per subgraph)", mode);
  if (num_subg == 0 || F == 0) return SG_OK;
  hipLaunchKernelGGL(pool_grad_table_kernel, dim3(num_subg + num_roots), dim3(64), 0, (hipStream_t)stream, d_dout, lddo, d_droots, lddr,
                     d_node_off, num_subg, d_rows, num_roots, n, F, mode, d_table, ldt);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_encode_codes(int kind, const void *d_src, uint32_t n, uint32_t dim, uint32_t *d_codes,
                               void *stream) {
  if (!d_src || !d_codes) return set_error(SG_ERR_INVALID, "sl_encode_codes: null argument");
  if (kind < 0 || kind > 2 || dim == 0 || dim > 32)
    return set_error(SG_ERR_INVALID, "sl_encode_codes: kind %d dim %u", kind, dim);
  if (n == 0) return SG_OK;
  hipLaunchKernelGGL(encode_codes_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, kind, d_src, n,
                     dim, d_codes);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_onehot_linear_fwd(const float *d_X, int64_t ldx, const uint32_t *d_codes, const float *d_Wt,
                                    const float *d_bias, uint32_t n, uint32_t F, uint32_t dim, float *d_out,
                                    int64_t ldo, void *stream) {
  if (!d_codes || !d_Wt || !d_out) return set_error(SG_ERR_INVALID, "sl_onehot_linear_fwd: null argument");
  if (dim == 0 || dim > (uint32_t)kMaxDim) return set_error(SG_ERR_INVALID, "sl_onehot_linear_fwd: dim %u (1..%d)", dim, kMaxDim);
  if (n == 0 || F == 0) return SG_OK;
  const uint64_t threads = (uint64_t)n * ((F + 3) / 4);
  hipLaunchKernelGGL(onehot_linear_fwd_kernel, dim3((uint32_t)((threads + kPB - 1) / kPB)), dim3(kPB), 0,
                     (hipStream_t)stream, d_X, ldx, d_codes, d_Wt, d_bias, n, F, dim, d_out, ldo);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_onehot_linear_bwd(const float *d_dout, int64_t lddo, const uint32_t *d_codes, uint32_t n, uint32_t F,
                                    uint32_t dim, float *d_dWt, float *d_dbias, float *d_partial,
                                    uint32_t partial_blocks, void *stream) {
  if (!d_dout || !d_codes || !d_dWt || !d_partial)
    return set_error(SG_ERR_INVALID, "sl_onehot_linear_bwd: null argument");
  if (dim == 0 || dim > (uint32_t)kMaxDim) return set_error(SG_ERR_INVALID, "sl_onehot_linear_bwd: dim %u (1..%d)", dim, kMaxDim);
  if (F == 0) return SG_OK;
  const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>(partial_blocks, (n + 63) / 64));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(onehot_linear_bwd_kernel, dim3(blocks), dim3(kPB), 0, st, d_dout, lddo, d_codes, n, F, dim,
                     d_partial);
  SHD_HIP(hipGetLastError());
  hipLaunchKernelGGL(onehot_linear_finish_kernel, dim3(((dim + 1) * F + kOhCols - 1) / kOhCols), dim3(kOhCols * kOhSlices), 0, st,
                     d_partial, blocks, F, dim, d_dWt, d_dbias);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}
