// aggregate.hip -- per-subgraph aggregation ops of the shaDow layers behind the
// sl_* C ABI (MI355X / gfx950).  All of them are HBM-bound gather / streaming
// kernels over the block-diagonal batch CSR produced by the sampler:
//
//   sl_gather_rows_f32   feat_full[subgs.node]               shaDow/minibatch.py:469
//   sl_csr_transpose     structure for the backward SpMM (A^T dY)
//   sl_degree_scales     D^-1 / D^-1/2 of adj_norm_rw / adj_norm_sym
//                        (frontend/graph_utils.py:67-145), optionally with a drop-edge mask
//   sl_spmm_csr_f32      Y = diag(rs) (A o W) diag(cs) X     shaDow/layers.py:326-327,433,475
//   sl_act_norm_fwd/bwd  sum_b norm_b(act(Z_b))              shaDow/layers.py:329-338,476-483
//
// Layout: rows are feature vectors of F fp32; one wavefront owns 64/LPR rows at a
// time, LPR = lanes per row (power of two), each lane moving float4 (16 B) so a
// 256-feature row is exactly one 1-KiB wave access.  MFMA is not used here: the
// dense feature x weight GEMMs stay on rocBLAS/hipBLASLt through torch.
#include <string.h>

#include <algorithm>

#include "common.h"
#include "actnorm_common.h"
#include "gat_act.h"

namespace shadow {

constexpr int kBlock = 256;

// Stores of kernels whose output the NEXT kernel reads (the gathered feature rows -> layer-0 aggregation, the aggregate
// -> the layer's GEMM): plain, so that the lines stay in the L2 / Infinity Cache for the consumer.  Round 2 had made them
// non-temporal together with the act_norm streams and lost 25 % on the F = 100 aggregation (102 -> 129 us) -- A/B per
// call site: scripts/ab_nt_stores.sh.  -DSHADOW_NT_GATHER_OUT=1 / -DSHADOW_NT_SPMM_OUT=1 rebuild the non-temporal forms.
#ifndef SHADOW_NT_GATHER_OUT
#define SHADOW_NT_GATHER_OUT 0
#endif
#ifndef SHADOW_NT_SPMM_OUT
#define SHADOW_NT_SPMM_OUT 0
#endif
#if SHADOW_NT_GATHER_OUT
#define SHD_ST_GATHER st4s
#else
#define SHD_ST_GATHER st4
#endif
#if SHADOW_NT_SPMM_OUT
#define SHD_ST_SPMM st4s
#else
#define SHD_ST_SPMM st4
#endif

// ---------------------------------------------------------------- gather
// out[i, :] = table[idx[i], :]; F % 4 == 0, rows 16-B aligned.
template <int LPR>
__global__ void gather_rows_kernel(const float *__restrict__ table, int64_t ld_table,
                                   const uint32_t *__restrict__ idx, float *__restrict__ out,
                                   int64_t ld_out, uint32_t n, uint32_t F) {
  const uint32_t rows_per_block = kBlock / LPR;
  const uint32_t sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  for (uint64_t r = (uint64_t)blockIdx.x * rows_per_block + sub; r < n;
       r += (uint64_t)gridDim.x * rows_per_block) {
    const float *src = table + (int64_t)idx[r] * ld_table;
    float *dst = out + (int64_t)r * ld_out;
    for (uint32_t c = l * 4; c < F; c += LPR * 4) st4(dst + c, ld4(src + c));
  }
}

__global__ void gather_rows_scalar_kernel(const float *__restrict__ table, int64_t ld_table,
                                          const uint32_t *__restrict__ idx, float *__restrict__ out,
                                          int64_t ld_out, uint32_t n, uint32_t F) {
  const uint64_t total = (uint64_t)n * F;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = t / F, c = t % F;
    out[r * ld_out + c] = table[(int64_t)idx[r] * ld_table + c];
  }
}

// ---------------------------------------------------------------- transpose
// Batch CSR -> transposed CSR (+ permutation into the original edge order), all
// edge-parallel.  Fast path: the structure is symmetric (undirected graphs with
// sorted rows, the normal case) => A^T has the same indptr/indices and the
// permutation is "position of the mate edge", found by binary search.  A
// device-side flag switches to the general counting path when a mate is missing.

// row of every edge (lower_bound over indptr)
__global__ void edge_rows_kernel(const uint32_t *__restrict__ indptr, uint32_t n, uint32_t e,
                                 uint32_t *__restrict__ edge_row) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < e; p += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t lo = 0, hi = n;           // largest i with indptr[i] <= p
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (indptr[mid] <= p) lo = mid; else hi = mid; }
    edge_row[p] = lo;
  }
}

__global__ void tr_sym_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                              const uint32_t *__restrict__ edge_row, uint32_t n, uint32_t e,
                              uint32_t *__restrict__ t_indptr, uint32_t *__restrict__ t_indices,
                              uint32_t *__restrict__ t_perm, uint32_t *__restrict__ flag) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = tid; i <= n; i += nt) t_indptr[i] = indptr[i];
  for (uint64_t p = tid; p < e; p += nt) {
    const uint32_t i = edge_row[p], j = indices[p];
    t_indices[p] = j;
    uint32_t lo = indptr[j], hi = indptr[j + 1];
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (indices[mid] < i) lo = mid + 1; else hi = mid; }
    if (lo < indptr[j + 1] && indices[lo] == i) t_perm[lo] = (uint32_t)p;   // entry (j, i) of A^T comes from edge p
    else atomicOr(flag, 1u);
  }
}

// every slot of t_perm must have been claimed exactly once
__global__ void tr_check_kernel(const uint32_t *__restrict__ t_perm, uint32_t e, uint32_t *__restrict__ flag) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < e; p += (uint64_t)gridDim.x * blockDim.x)
    if (t_perm[p] == 0xFFFFFFFFu) atomicOr(flag, 1u);
}

// ---- general path (runs only when *flag != 0)
__global__ void tr_count_kernel(const uint32_t *__restrict__ indices, uint32_t n, uint32_t e,
                                uint32_t *__restrict__ cnt, const uint32_t *__restrict__ flag) {
  if (*flag == 0) return;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = tid; t < e; t += nt) atomicAdd(&cnt[indices[t] + 1], 1u);
}

__global__ void tr_zero_kernel(uint32_t *__restrict__ a, uint32_t n1, const uint32_t *__restrict__ flag) {
  if (*flag == 0) return;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n1; t += (uint64_t)gridDim.x * blockDim.x) a[t] = 0;
}

// single-workgroup inclusive scan in place over a[0..n1) (a[i+1] = count of row i, a[0] = 0)
__global__ void tr_scan_kernel(uint32_t *__restrict__ a, uint32_t n1, uint32_t *__restrict__ cursor,
                               const uint32_t *__restrict__ flag) {
  if (*flag == 0) return;
  __shared__ uint32_t wsum[32];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n1; base += blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n1 ? a[i] : 0u;
    const uint32_t incl = wave_incl_scan(v);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (uint32_t w = 0; w < (blockDim.x >> 6); w++) { const uint32_t x = wsum[w]; if (w < wave_id()) pre += x; tot += x; }
    const uint32_t c = carry_s;
    if (i < n1) { a[i] = c + pre + incl; cursor[i] = c + pre + incl; }
    __syncthreads();
    if (threadIdx.x == 0) carry_s = c + tot;
    __syncthreads();
  }
}

__global__ void tr_fill_kernel(const uint32_t *__restrict__ indices, const uint32_t *__restrict__ edge_row,
                               uint32_t e, uint32_t *__restrict__ cursor, uint32_t *__restrict__ t_indices,
                               uint32_t *__restrict__ t_perm, const uint32_t *__restrict__ flag) {
  if (*flag == 0) return;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < e; p += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t q = atomicAdd(&cursor[indices[p]], 1u);
    t_indices[q] = edge_row[p];
    t_perm[q] = (uint32_t)p;
  }
}

// Order every transposed row by edge position (== by source row): one wavefront
// per row, rank sort through registers (rows are short; a hub row costs k^2/64).
__global__ void tr_sort_rows_kernel(const uint32_t *__restrict__ t_indptr, uint32_t n,
                                    uint32_t *__restrict__ t_indices, uint32_t *__restrict__ t_perm,
                                    uint32_t *__restrict__ tmp_idx, uint32_t *__restrict__ tmp_perm,
                                    const uint32_t *__restrict__ flag) {
  if (*flag == 0) return;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = lane_id();
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint64_t j = wave; j < n; j += nwaves) {
    const uint32_t a = t_indptr[j], b = t_indptr[j + 1];
    if (b - a < 2) continue;
    for (uint32_t x = a + lane; x < b; x += 64) {
      const uint32_t kp = t_perm[x];
      uint32_t r = 0;
      for (uint32_t y = a; y < b; y++) r += (t_perm[y] < kp) ? 1u : 0u;
      tmp_perm[a + r] = kp; tmp_idx[a + r] = t_indices[x];
    }
  }
}

__global__ void tr_copy_back_kernel(const uint32_t *__restrict__ t_indptr, uint32_t n, uint32_t e,
                                    uint32_t *__restrict__ t_indices, uint32_t *__restrict__ t_perm,
                                    const uint32_t *__restrict__ tmp_idx, const uint32_t *__restrict__ tmp_perm,
                                    const uint32_t *__restrict__ flag) {
  if (*flag == 0) return;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < e; p += (uint64_t)gridDim.x * blockDim.x) {
    // rows with a single entry were not touched by the sort
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (t_indptr[mid] <= p) lo = mid; else hi = mid; }
    if (t_indptr[lo + 1] - t_indptr[lo] >= 2) { t_indices[p] = tmp_idx[p]; t_perm[p] = tmp_perm[p]; }
  }
}

// ---------------------------------------------------------------- degree scales
// mode 0 (rw):  row_scale[i] = 1 / max(1, sum_j w_ij)          graph_utils.py:84-94
// mode 1 (sym): row_scale[i] = max(1, sum_j w_ij)^-1/2         graph_utils.py:140-142
__global__ void degree_scales_kernel(const uint32_t *__restrict__ indptr, const float *__restrict__ edge_w,
                                     uint32_t n, int mode, float *__restrict__ row_scale) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t a = indptr[i], b = indptr[i + 1];
    float d;
    if (edge_w) { d = 0.f; for (uint32_t p = a; p < b; p++) d += edge_w[p]; }
    else d = (float)(b - a);
    d = fmaxf(d, 1.0f);
    row_scale[i] = mode == 0 ? 1.0f / d : 1.0f / sqrtf(d);
  }
}

// ---------------------------------------------------------------- SpMM
// Y[i,:] = rs[i] * sum_{p in row i} w[perm[p]] * cs[col_p] * X[col_p,:]
// LPR lanes per row (F/4 <= LPR*CH), each lane CH float4 chunks.
template <int LPR, int CH>
__global__ void spmm_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                            const float *__restrict__ edge_w, const uint32_t *__restrict__ edge_perm,
                            const float *__restrict__ row_scale, const float *__restrict__ col_scale,
                            const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy,
                            uint32_t n, uint32_t F) {
  const uint32_t rows_per_block = kBlock / LPR;
  const uint32_t sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  // XCD-aware order: consecutive logical blocks (rows of the same subgraphs) share an XCD's L2
  const uint32_t nb = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t per = nb >> 3;          // the grid is a multiple of 8 blocks
  const uint32_t lb = xcd * per + slot;
  const uint64_t r = (uint64_t)lb * rows_per_block + sub;
  if (r >= n) return;
  float4 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t a = indptr[r], b = indptr[r + 1];
  uint32_t p = a;
  // two edges per iteration: both gathers in flight
  for (; p + 1 < b; p += 2) {
    const uint32_t c0 = indices[p], c1 = indices[p + 1];
    float w0 = 1.f, w1 = 1.f;
    if (edge_w) { w0 = edge_w[edge_perm ? edge_perm[p] : p]; w1 = edge_w[edge_perm ? edge_perm[p + 1] : p + 1]; }
    if (col_scale) { w0 *= col_scale[c0]; w1 *= col_scale[c1]; }
    const float *x0 = X + (int64_t)c0 * ldx, *x1 = X + (int64_t)c1 * ldx;
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const uint32_t f = (l + c * LPR) * 4;
      if (f < F) {
        const float4 v0 = ld4(x0 + f), v1 = ld4(x1 + f);
        acc[c].x += w0 * v0.x; acc[c].y += w0 * v0.y; acc[c].z += w0 * v0.z; acc[c].w += w0 * v0.w;
        acc[c].x += w1 * v1.x; acc[c].y += w1 * v1.y; acc[c].z += w1 * v1.z; acc[c].w += w1 * v1.w;
      }
    }
  }
  if (p < b) {
    const uint32_t c0 = indices[p];
    float w0 = 1.f;
    if (edge_w) w0 = edge_w[edge_perm ? edge_perm[p] : p];
    if (col_scale) w0 *= col_scale[c0];
    const float *x0 = X + (int64_t)c0 * ldx;
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const uint32_t f = (l + c * LPR) * 4;
      if (f < F) {
        const float4 v0 = ld4(x0 + f);
        acc[c].x += w0 * v0.x; acc[c].y += w0 * v0.y; acc[c].z += w0 * v0.z; acc[c].w += w0 * v0.w;
      }
    }
  }
  const float rs = row_scale ? row_scale[r] : 1.0f;
  float *y = Y + (int64_t)r * ldy;
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const uint32_t f = (l + c * LPR) * 4;
    if (f < F) st4(y + f, make_float4(acc[c].x * rs, acc[c].y * rs, acc[c].z * rs, acc[c].w * rs));
  }
}

__device__ __forceinline__ uint32_t rl_u32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ float rl_f32(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// ---------------------------------------------------------------------------
// Persistent, software-pipelined multi-row SpMM.  A "unit" = LPG lanes (a whole wavefront for
// 128 < F <= 256) walks a contiguous chunk of row groups (R rows each).  The
// three dependent HBM round trips of a group (row pointers -> edge ids [-> edge weights /
// column scales] -> feature rows) are spread over the pipeline stages of four consecutive
// groups, so each loop iteration waits for exactly one round trip -- the feature-row gathers
// of the current group -- while the pointer / id loads of the next three groups ride along.
// Subgraph rows are tiny (~2 edges): the R+1 row pointers and all edge ids of a group sit one
// per lane, column ids move to scalar registers (v_readlane) and up to KMAX 1-KiB feature-row
// gathers are issued back to back; groups with more edges fall back to a streaming loop.
// Contiguous chunks keep a subgraph's rows (which gather each other's features) on one CU.
// ---------------------------------------------------------------------------
template <int LPG>
__device__ __forceinline__ uint32_t grp_u32(uint32_t v, int k, int gb) {
  if constexpr (LPG == 64) return rl_u32(v, k);
  else return (uint32_t)__builtin_amdgcn_ds_bpermute((gb + k) << 2, (int)v);
}
template <int LPG>
__device__ __forceinline__ float grp_f32(float v, int k, int gb) {
  return __int_as_float((int)grp_u32<LPG>((uint32_t)__float_as_int(v), k, gb));
}

template <int R, int KMAX, int LPG>
__global__ void __launch_bounds__(kBlock)
spmm_pipe_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                 const float *__restrict__ edge_w, const uint32_t *__restrict__ edge_perm,
                 const float *__restrict__ row_scale, const float *__restrict__ col_scale,
                 const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy,
                 uint32_t n, uint32_t F, uint32_t chunk, float *__restrict__ row_amax, uint32_t amax_join) {
  static_assert(KMAX <= LPG && R + 1 <= LPG, "edge ids / row pointers live one per lane");
  const uint32_t lane = lane_id();
  const uint32_t sl = lane & (LPG - 1);              // lane inside the unit
  const int gb = (int)(lane - sl);                   // first lane of the unit
  // XCD-aware block order (the grid is a multiple of 8 blocks)
  const uint32_t nb = gridDim.x, per = nb >> 3;
  const uint32_t lb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  uint64_t unit = ((uint64_t)lb * (kBlock / 64) + wave_id()) * (64 / LPG) + lane / LPG;
  if (LPG == 64) unit = (uint64_t)__builtin_amdgcn_readfirstlane((int)unit);      // (units < 2^31)
  const uint64_t G = ((uint64_t)n + R - 1) / R;
  const uint64_t g0 = unit * chunk;
  if (__ballot(g0 < G) == 0) return;
  const uint64_t g1 = min(G, g0 + chunk);
  const uint32_t f = sl * 4;
  const bool on = f < F;
  const uint32_t ipn = indptr[n];
  uint32_t ipl_b = ipn, ipl_c = ipn, ipl_d = ipn;    // row pointers of groups it+2, it+1, it
  uint32_t col_c = 0, pe_c = 0, col_d = 0;
  float w_d = 1.0f;
  for (int64_t it = -3; it < (int64_t)chunk; it++) {
    // ---- stage A: row pointers of group it+3
    uint32_t ipl_a;
    {
      const uint64_t g = g0 + (uint64_t)(it + 3);
      const uint64_t ra = (g < g1) ? g * R : (uint64_t)n;
      ipl_a = indptr[min(ra + sl, (uint64_t)n)];     // lanes past the last row hold indptr[n]
    }
    // ---- stage B1: edge ids (and permuted edge positions) of group it+2
    uint32_t col_b = 0, pe_b = 0;
    {
      const uint32_t e0 = grp_u32<LPG>(ipl_b, 0, gb), E = grp_u32<LPG>(ipl_b, R, gb) - e0;
      if (sl < min(E, (uint32_t)KMAX)) {
        col_b = indices[e0 + sl];
        pe_b = edge_perm ? edge_perm[e0 + sl] : e0 + sl;
      }
    }
    // ---- stage B2: edge weights / column scales of group it+1
    float w_c = 1.0f;
    if (edge_w || col_scale) {
      const uint32_t e0 = grp_u32<LPG>(ipl_c, 0, gb), E = grp_u32<LPG>(ipl_c, R, gb) - e0;
      if (sl < min(E, (uint32_t)KMAX)) {
        if (edge_w) w_c = edge_w[pe_c];
        if (col_scale) w_c *= col_scale[col_c];
      }
    }
    // ---- stage C: gather, accumulate and store group it
    {
      const uint64_t g = g0 + (uint64_t)(it < 0 ? 0 : it);
      const uint64_t r0 = (it >= 0 && g < g1) ? g * R : (uint64_t)n;
      uint32_t ip[R + 1];
#pragma unroll
      for (int q = 0; q <= R; q++) ip[q] = grp_u32<LPG>(ipl_d, q, gb);
      const uint32_t e0 = ip[0], E = ip[R] - e0;
      float rsl = 1.0f;
      if (row_scale && r0 + sl < n && sl < R) rsl = row_scale[r0 + sl];
      float4 acc[R];
#pragma unroll
      for (int q = 0; q < R; q++) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (E <= (uint32_t)KMAX) {
        float4 x[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
          x[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((uint32_t)k < E) {
            const uint32_t c = grp_u32<LPG>(col_d, k, gb);
            if (on) x[k] = ld4(X + (int64_t)c * ldx + f);
          }
        }
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
          if ((uint32_t)k < E) {
            const float wk = grp_f32<LPG>(w_d, k, gb);
            const uint32_t ek = e0 + k;
#pragma unroll
            for (int q = 0; q < R; q++) {
              if (ek >= ip[q] && ek < ip[q + 1]) {
                acc[q].x += wk * x[k].x; acc[q].y += wk * x[k].y; acc[q].z += wk * x[k].z; acc[q].w += wk * x[k].w;
              }
            }
          }
        }
      } else {
        // long rows (hubs): stream the edges of each row, two gathers in flight
#pragma unroll
        for (int q = 0; q < R; q++) {
          uint32_t p = ip[q];
          const uint32_t b = ip[q + 1];
#ifndef SHADOW_SPMM_LONG
#define SHADOW_SPMM_LONG 6      // gathers in flight on the long rows before the pairs (scripts/micro/ab_spmm_long.sh)
#endif
#if SHADOW_SPMM_LONG > 2
          for (; p + (SHADOW_SPMM_LONG - 1) < b; p += SHADOW_SPMM_LONG) {
            uint32_t c[SHADOW_SPMM_LONG];
            float w[SHADOW_SPMM_LONG];
#pragma unroll
            for (int k = 0; k < SHADOW_SPMM_LONG; k++) c[k] = indices[p + k];
#pragma unroll
            for (int k = 0; k < SHADOW_SPMM_LONG; k++) {
              w[k] = 1.f;
              if (edge_w) w[k] = edge_w[edge_perm ? edge_perm[p + k] : p + k];
              if (col_scale) w[k] *= col_scale[c[k]];
            }
            if (on) {
              float4 v[SHADOW_SPMM_LONG];
#pragma unroll
              for (int k = 0; k < SHADOW_SPMM_LONG; k++) v[k] = ld4(X + (int64_t)c[k] * ldx + f);
#pragma unroll
              for (int k = 0; k < SHADOW_SPMM_LONG; k++) { acc[q].x += w[k] * v[k].x; acc[q].y += w[k] * v[k].y; acc[q].z += w[k] * v[k].z; acc[q].w += w[k] * v[k].w; }
            }
          }
#endif
          for (; p + 1 < b; p += 2) {
            const uint32_t c0 = indices[p], c1 = indices[p + 1];
            float w0 = 1.f, w1 = 1.f;
            if (edge_w) { w0 = edge_w[edge_perm ? edge_perm[p] : p]; w1 = edge_w[edge_perm ? edge_perm[p + 1] : p + 1]; }
            if (col_scale) { w0 *= col_scale[c0]; w1 *= col_scale[c1]; }
            if (on) {
              const float4 v0 = ld4(X + (int64_t)c0 * ldx + f), v1 = ld4(X + (int64_t)c1 * ldx + f);
              acc[q].x += w0 * v0.x; acc[q].y += w0 * v0.y; acc[q].z += w0 * v0.z; acc[q].w += w0 * v0.w;
              acc[q].x += w1 * v1.x; acc[q].y += w1 * v1.y; acc[q].z += w1 * v1.z; acc[q].w += w1 * v1.w;
            }
          }
          if (p < b) {
            const uint32_t c0 = indices[p];
            float w0 = 1.f;
            if (edge_w) w0 = edge_w[edge_perm ? edge_perm[p] : p];
            if (col_scale) w0 *= col_scale[c0];
            if (on) {
              const float4 v0 = ld4(X + (int64_t)c0 * ldx + f);
              acc[q].x += w0 * v0.x; acc[q].y += w0 * v0.y; acc[q].z += w0 * v0.z; acc[q].w += w0 * v0.w;
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < R; q++) {
        const float rs = grp_f32<LPG>(rsl, q, gb);
        const float4 y = make_float4(acc[q].x * rs, acc[q].y * rs, acc[q].z * rs, acc[q].w * rs);
        if (r0 + q < n && on) st4(Y + (int64_t)(r0 + q) * ldy + f, y);
        if (row_amax) {                   // (largest magnitude of the row: the fp16 operand scale of the GEMM that reads Y)
          const float m = group_max<LPG>(on ? amax4(y) : 0.f);
          // (amax_join: the array holds the maxima of other columns of the same operand -- one wavefront owns the row)
          if (sl == 0 && r0 + q < n) row_amax[r0 + q] = amax_join ? fmaxf(row_amax[r0 + q], m) : m;
        }
      }
    }
    ipl_d = ipl_c; ipl_c = ipl_b; ipl_b = ipl_a;
    col_d = col_c; w_d = w_c; col_c = col_b; pe_c = pe_b;
  }
}

// ---------------------------------------------------------------------------
// Block-diagonal SpMM with the subgraph's feature tile staged in LDS.  A minibatch adjacency is
// block diagonal (graph.py:280-330): the rows of subgraph s only gather feature rows of subgraph s,
// each ~2x.  One workgroup takes (subgraph, 32-float column tile): the tile of X (n_s x 128 B), the
// subgraph's row pointers, its column ids and final edge weights are loaded with coalesced,
// mutually independent loads (one HBM round trip), then every edge is an LDS read.  Traffic
// through L2 drops from (e + n) to 2n feature rows.  Subgraphs that do not fit the LDS budget
// gather from global memory in the same kernel.
// ---------------------------------------------------------------------------
constexpr int kBdTile = 32;            // floats per column tile (8 lanes x float4)
constexpr int kBdBlock = 1024;         // threads per workgroup (128 row groups of 8 lanes)
constexpr int kBdRowPad = kBdTile + 4; // LDS row stride in floats (16 B pad rotates the banks)
constexpr int kBdMaxRows = 384;        // rows of one subgraph tile held in LDS (12 per 8-lane group)
constexpr int kBdMaxEdges = 1024;      // edges of one subgraph held in LDS (4 per thread)
constexpr int kBdKX = kBdMaxRows / (kBdBlock / 8);
constexpr int kBdKE = kBdMaxEdges / kBdBlock;

struct BdItem { uint32_t a, ns, e0, es, t, first, valid; };

// Layer-0 fusion (shaDow/minibatch.py:469 + the layer's input nn.Dropout, layers.py:430,471): row i of X is
// table[ids[i]] (the feature gather), optionally dropped out with the counter-hash mask of act_norm (same rule:
// include/shadow_hip.h), and the staged rows are also written out densely (xout) for the layer's other consumers.
struct BdGather {
  const float *table;       // nullptr: X is dense
  int64_t ldt;
  const uint32_t *ids;
  float *xout;              // [n, F] gathered (+ dropped) copy, or nullptr
  int64_t ldxo;
  uint32_t drop_thr;        // 0: no dropout
  float drop_scale;
  uint32_t seed_lo, seed_hi;
  uint32_t zero_id;         // ids[r] >= zero_id: the row is all zeros and is not fetched (row-mapped inputs); 0xFFFFFFFF: none
};

// the four features f..f+3 of batch row r after the input dropout (mask rule: actnorm_common.h)
__device__ __forceinline__ float4 bd_drop4(const BdGather &g, float4 v, uint64_t r, uint32_t f) {
  if (!g.drop_thr) return v;
  const uint32_t keep = drop_keep4_raw(g.seed_lo, g.seed_hi, g.drop_thr, r, f);
  v.x = (keep & 1u) ? v.x * g.drop_scale : 0.f;
  v.y = (keep & 2u) ? v.y * g.drop_scale : 0.f;
  v.z = (keep & 4u) ? v.z * g.drop_scale : 0.f;
  v.w = (keep & 8u) ? v.w * g.drop_scale : 0.f;
  return v;
}

// row r of the (virtual) input matrix: dense X or the gathered, dropped table row
// kGather: 0 dense X; 1 table[ids[r]] with the dropout hash (+ the dense copy xout); 2 table[ids[r]] through a plain row map
// (rows mapped to zero_id are zeros and not fetched; no hash, no copy: the lean instantiation of sl_spmm_blockdiag_rows_f32 --
// the full one spills 34 VGPRs at the kernel's 64-register cap)
template <int kGather>
__device__ __forceinline__ float4 bd_load4(const BdGather &g, const float *__restrict__ X, int64_t ldx, uint64_t r, uint32_t f) {
  if (kGather == 0) return ld4(X + (int64_t)r * ldx + f);
  const uint32_t id = g.ids[r];
  if (kGather == 2) return id >= g.zero_id ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(g.table + (int64_t)id * g.ldt + f);
  return bd_drop4(g, ld4(g.table + (int64_t)id * g.ldt + f), r, f);
}

// out[i, 0:F] = dropout(table[idx[i], :]), out[i, F:Fpad] = 0: the feature gather of a batch (shaDow/minibatch.py:469)
// with layer 0's input dropout (layers.py:430,471) in the same pass, written into rows padded to whole 128-byte
// lines so that every later reader (SpMM tiles, GEMM operand loads) works on aligned rows.
template <int LPR>
__global__ void gather_rows_drop_kernel(const float *__restrict__ table, int64_t ld_table, const uint32_t *__restrict__ idx,
                                        float *__restrict__ out, int64_t ld_out, uint32_t n, uint32_t F, uint32_t Fpad,
                                        BdGather g, float *__restrict__ row_amax) {
  const uint32_t rows_per_block = kBlock / LPR;
  const uint32_t sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  // (the trip count is rounded up to whole lane groups' worth of rows: the row maximum below is a DPP reduction)
  const uint64_t stride = (uint64_t)gridDim.x * rows_per_block;
  for (uint64_t r0 = (uint64_t)blockIdx.x * rows_per_block; r0 < n; r0 += stride) {
    const uint64_t r = r0 + sub;
    float mx = 0.f;
    if (r < n) {
      const float *src = table + (int64_t)idx[r] * ld_table;
      float *dst = out + (int64_t)r * ld_out;
      for (uint32_t c = l * 4; c < Fpad; c += LPR * 4) {
        const float4 v = c < F ? bd_drop4(g, ld4(src + c), r, c) : make_float4(0.f, 0.f, 0.f, 0.f);
        SHD_ST_GATHER(dst + c, v);
        mx = fmaxf(mx, amax4(v));
      }
    }
    if (row_amax) {                       // largest magnitude of the row: the fp16 operand scale of the GEMM that reads it
      mx = group_max<LPR>(mx);
      if (l == 0 && r < n) row_amax[r] = mx;
    }
  }
}

// float4 column of lane l8 in tile t, every tile at most 8 wide.  Rows on a 128-byte pitch (``lines``: the padded rows of
// LazyRows.gather_dropped and the product that keeps their pitch): tile t is the t-th LINE of the row -- F = 100:
// 8+8+8+1 -- so that every line is fetched and written by exactly one work item.  Otherwise the F/4 float4 columns are
// split evenly over the tiles (F = 100: 7+6+6+6).
__device__ __forceinline__ uint32_t bd_col4(uint32_t t, uint32_t tiles, uint32_t nf4, uint32_t l8, bool *on, uint32_t lines) {
  const uint32_t c0 = lines ? 8u * t : t * nf4 / tiles, c1 = lines ? min(nf4, 8u * t + 8u) : (t + 1u) * nf4 / tiles;
  *on = c0 + l8 < c1;
  return (c0 + l8) * 4u;
}

// Work items: (subgraph, column tile).  A workgroup takes the tiles of a tile GROUP (tg consecutive tiles of one
// subgraph) back to back: the subgraph's row pointers / column ids / weights are staged once per group, and the
// slices of a feature row that the group's tiles read follow each other on the same CU (rows of 100 floats are not
// line aligned: neighbouring tiles share sectors, which now hit L1 / L2 instead of being fetched again).
// k-th item of workgroup b: group gi = b + (k / tg) * gridDim.x, tile (gi % groups) * tg + k % tg of subgraph gi / groups.
// Headers are wave-uniform (scalar loads).
__device__ __forceinline__ BdItem bd_item(uint32_t k, uint32_t tg, uint32_t groups, uint32_t tiles, uint32_t P,
                                          const uint32_t *__restrict__ node_off, const uint32_t *__restrict__ edge_off) {
  BdItem h; h.a = 0; h.ns = 0; h.e0 = 0; h.es = 0; h.t = 0; h.first = 0; h.valid = 0;
  const uint64_t gi = (uint64_t)blockIdx.x + (uint64_t)(k / tg) * gridDim.x;
  if (gi < (uint64_t)P * groups) {
    const uint32_t s = (uint32_t)(gi / groups);
    h.t = (uint32_t)(gi % groups) * tg + k % tg;
    h.first = (k % tg) == 0;
    h.valid = h.t < tiles;
    h.a = node_off[s]; h.ns = node_off[s + 1] - h.a;
    h.e0 = edge_off[s]; h.es = edge_off[s + 1] - h.e0;
  }
  return h;
}

// Largest magnitude of an output row for the fp16 operand scale of the GEMM that reads it (gemm_common.h).  A row's tiles
// are produced by different work items: every (row, tile) folds its 8 lanes and joins with an atomic maximum on the bit
// pattern (magnitudes are non-negative floats: integer order = float order; the result does not depend on the order).
// The caller zeroes the array -- or pre-fills it with the maxima of other columns of the same operand.
__device__ __forceinline__ void bd_row_amax(uint32_t *dst, float m, uint32_t l8) {
  m = group_max<8>(m);
  if (l8 == 0 && m > 0.f) atomicMax(dst, __float_as_uint(m));
}

// (two 1024-thread workgroups per CU = 8 wavefronts per SIMD: at most 64 VGPRs)
template <int kGather>
__global__ void __launch_bounds__(kBdBlock, 8)
spmm_blockdiag_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                      const float *__restrict__ edge_w, const uint32_t *__restrict__ edge_perm,
                      const float *__restrict__ row_scale, const float *__restrict__ col_scale,
                      const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy,
                      uint32_t F, const uint32_t *__restrict__ node_off, const uint32_t *__restrict__ edge_off,
                      uint32_t P, uint32_t tiles, uint32_t tg, uint32_t cap_rows, BdGather g, uint32_t *__restrict__ amax_bits,
                      uint32_t lines) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem[];
  float *xs = reinterpret_cast<float *>(bd_smem);                        // [cap_rows][kBdRowPad]
  uint32_t *ips = reinterpret_cast<uint32_t *>(xs + (size_t)cap_rows * kBdRowPad);   // [cap_rows + 4]
  uint32_t *cols = ips + cap_rows + 4;                                   // [kBdMaxEdges] local column ids
  float *ws = reinterpret_cast<float *>(cols + kBdMaxEdges);             // [kBdMaxEdges] final edge weights
  const uint32_t tid = threadIdx.x;
  const uint32_t l8 = tid & 7u, rg = tid >> 3;                           // 8 lanes per row, 32 rows per pass
  const bool weighted = (edge_w != nullptr) || (col_scale != nullptr);
  const uint32_t groups = (tiles + tg - 1) / tg;
  // ---- software pipeline over this workgroup's items: the global loads of item k+1 are issued
  //      into registers before item k is computed out of LDS
  float4 xv[kBdKX];
  float rsv[kBdKX], rsc[kBdKX];      // row scales of the prefetched / current item
  uint32_t iv[2], cv[kBdKE];
  float wv[kBdKE];
  auto fits = [&](const BdItem &h) { return h.ns != 0 && h.ns <= cap_rows && h.es <= (uint32_t)kBdMaxEdges; };
  auto prefetch = [&](const BdItem &h) {
    if (!h.valid || !fits(h)) return;
    bool pon;
    const uint32_t f = bd_col4(h.t, tiles, F >> 2, l8, &pon, lines);
#pragma unroll
    for (int k = 0; k < kBdKX; k++) {
      const uint32_t i = rg + k * (kBdBlock / 8);
      xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      rsv[k] = 1.0f;
      if (i < h.ns && pon) {
        xv[k] = bd_load4<kGather>(g, X, ldx, (uint64_t)h.a + i, f);
        if (kGather == 1 && g.xout) st4(g.xout + (int64_t)(h.a + i) * g.ldxo + f, xv[k]);       // the dense copy of the staged rows
      }
      if (i < h.ns && row_scale) rsv[k] = row_scale[h.a + i];
    }
    if (!h.first) return;                 // the group's first tile staged the subgraph's structure already
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t i = tid + k * kBdBlock;
      iv[k] = (i <= h.ns) ? indptr[h.a + i] - h.e0 : 0u;
    }
#pragma unroll
    for (int k = 0; k < kBdKE; k++) {
      const uint32_t p = tid + k * kBdBlock;
      cv[k] = 0; wv[k] = 1.0f;
      if (p < h.es) {
        const uint32_t c = indices[h.e0 + p];
        cv[k] = c - h.a;
        if (edge_w) wv[k] = edge_w[edge_perm ? edge_perm[h.e0 + p] : h.e0 + p];
        if (col_scale) wv[k] *= col_scale[c];
      }
    }
  };
  uint32_t kq = 0;
  BdItem cur = bd_item(0, tg, groups, tiles, P, node_off, edge_off);
  BdItem nxt = bd_item(1, tg, groups, tiles, P, node_off, edge_off);
  prefetch(cur);
  for (;; kq++) {
    if ((uint64_t)blockIdx.x + (uint64_t)(kq / tg) * gridDim.x >= (uint64_t)P * groups) break;   // (wave-uniform)
    bool on;
    const uint32_t f = bd_col4(cur.t, tiles, F >> 2, l8, &on, lines);
    on = on && cur.valid;
    const bool in_lds = cur.valid && fits(cur);
    __syncthreads();                                                     // the previous item's readers are done
    if (in_lds) {
#pragma unroll
      for (int k = 0; k < kBdKX; k++) {
        const uint32_t i = rg + k * (kBdBlock / 8);
        if (i < cur.ns) *reinterpret_cast<float4 *>(xs + (size_t)i * kBdRowPad + l8 * 4) = xv[k];
        rsc[k] = rsv[k];
      }
      if (cur.first) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const uint32_t i = tid + k * kBdBlock;
          if (i <= cur.ns) ips[i] = iv[k];
        }
#pragma unroll
        for (int k = 0; k < kBdKE; k++) {
          const uint32_t p = tid + k * kBdBlock;
          if (p < cur.es) { cols[p] = cv[k]; if (weighted) ws[p] = wv[k]; }
        }
      }
    }
    __syncthreads();
    const BdItem nn = bd_item(kq + 2, tg, groups, tiles, P, node_off, edge_off);
    prefetch(nxt);
    if (in_lds) {
#pragma unroll
      for (int k = 0; k < kBdKX; k++) {
        const uint32_t i = rg + k * (kBdBlock / 8);
        if (i < cur.ns) {
          const uint32_t p0 = ips[i], p1 = ips[i + 1];
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          for (uint32_t p = p0; p < p1; p++) {
            const float4 v = *reinterpret_cast<const float4 *>(xs + (size_t)cols[p] * kBdRowPad + l8 * 4);
            const float we = weighted ? ws[p] : 1.0f;
            acc.x += we * v.x; acc.y += we * v.y; acc.z += we * v.z; acc.w += we * v.w;
          }
          const float rs = rsc[k];
          const float4 y = make_float4(acc.x * rs, acc.y * rs, acc.z * rs, acc.w * rs);
          if (on) SHD_ST_SPMM(Y + (int64_t)(cur.a + i) * ldy + f, y);
          else if (lines == 2 && cur.valid) SHD_ST_SPMM(Y + (int64_t)(cur.a + i) * ldy + f, make_float4(0.f, 0.f, 0.f, 0.f));   // the pad of the row's last line
          if (amax_bits) bd_row_amax(amax_bits + cur.a + i, on ? amax4(y) : 0.f, l8);
        }
      }
    } else if (cur.valid) {
      // oversize subgraph: gather the feature rows from global memory
      for (uint32_t i = rg; i < cur.ns; i += kBdBlock / 8) {
        if (kGather == 1 && g.xout && on) st4(g.xout + (int64_t)(cur.a + i) * g.ldxo + f, bd_load4<kGather>(g, X, ldx, (uint64_t)cur.a + i, f));
        const uint32_t p0 = indptr[cur.a + i], p1 = indptr[cur.a + i + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t p = p0; p < p1; p++) {
          const uint32_t c = indices[p];
          float we = 1.0f;
          if (edge_w) we = edge_w[edge_perm ? edge_perm[p] : p];
          if (col_scale) we *= col_scale[c];
          if (on) {
            const float4 v = bd_load4<kGather>(g, X, ldx, (uint64_t)c, f);
            acc.x += we * v.x; acc.y += we * v.y; acc.z += we * v.z; acc.w += we * v.w;
          }
        }
        const float rs = row_scale ? row_scale[cur.a + i] : 1.0f;
        const float4 y = make_float4(acc.x * rs, acc.y * rs, acc.z * rs, acc.w * rs);
        if (on) SHD_ST_SPMM(Y + (int64_t)(cur.a + i) * ldy + f, y);
        else if (lines == 2) SHD_ST_SPMM(Y + (int64_t)(cur.a + i) * ldy + f, make_float4(0.f, 0.f, 0.f, 0.f));
        if (amax_bits) bd_row_amax(amax_bits + cur.a + i, on ? amax4(y) : 0.f, l8);
      }
    }
    cur = nxt; nxt = nn;
  }
}

// scalar fallback for F % 4 != 0
__global__ void spmm_scalar_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                   const float *__restrict__ edge_w, const uint32_t *__restrict__ edge_perm,
                                   const float *__restrict__ row_scale, const float *__restrict__ col_scale,
                                   const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy,
                                   uint32_t n, uint32_t F) {
  const uint64_t total = (uint64_t)n * F;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = t / F, f = t % F;
    float acc = 0.f;
    for (uint32_t p = indptr[r]; p < indptr[r + 1]; p++) {
      const uint32_t c = indices[p];
      float w = edge_w ? edge_w[edge_perm ? edge_perm[p] : p] : 1.0f;
      if (col_scale) w *= col_scale[c];
      acc += w * X[(int64_t)c * ldx + f];
    }
    Y[r * ldy + f] = acc * (row_scale ? row_scale[r] : 1.0f);
  }
}

// ---------------------------------------------------------------- act + row norm
struct ActNormParams {
  const float *bias[2];    // optional per-branch bias [F] added to Z before the activation (fused Linear bias)
  float *dbias;            // [nb, F] bias gradients (backward, optional)
  const float *Z[2];       // branch inputs [n, F]
  int64_t ldz[2];
  int act[2];              // activation applied to the branch input
  const float *scale;      // [nb, F]
  const float *offset;     // [nb, F]
  int nb;                  // 1 or 2 branches
  uint32_t n, F, seg;      // seg = normalisation segment (F, or the head slice for GAT)
  float out_scale;         // 1 (GCN/SAGE: sum), 0.5 (GAT: (self+neigh)/2)
  float eps;               // 1e-9
  float *out; int64_t ldo; // forward output
  // backward
  const float *dout; int64_t lddo;
  float *dZ[2]; int64_t lddz[2];
  float *dscale;           // [nb, F]
  float *doffset;          // [nb, F]
  float *partial;          // [grid, nb, 3, F] per-block partial sums of dscale / doffset / dbias
  // fused dropout of the OUTPUT (= the next layer's input dropout, layers.py:430,471,601): element
  // (row r, column c) is kept iff  mix32(mix32(r_lo ^ seed_lo) + r_hi + seed_hi + c * 0x9E3779B1) >= drop_thr,
  // mix32 = the murmur3 finaliser; kept values are scaled by drop_scale = 1 / (1 - p).  drop_thr == 0: none.
  uint32_t drop_thr;
  float drop_scale;
  uint32_t seed_lo, seed_hi;
  // dual mode (something else reads the un-dropped output too, e.g. a residue / pooling read-out):
  // forward writes out (plain) AND out2 (dropped); backward adds the two incoming gradients,
  // dout (may be NULL) for the plain output and dout2 through the dropout mask
  float *out2; int64_t ldo2;
  const float *dout2; int64_t lddo2;
  // backward, optional: max_k |dZ[0][r, k]| per row (the fp16 operand scale of the GEMM that reads dZ[0] next, sl_row_amax)
  float *dz0_amax;
  int amax_branch;               // the branch whose dZ row maxima dz0_amax receives (0 unless the caller says otherwise)
  // forward, optional: the same for the output the next layer reads (out2 in dual mode, out otherwise)
  float *out_amax;
  // backward, optional: the output gradient is given for `n` SELECTED rows only (dout / dout2 are [n, F] compact, row i of
  // them belongs to row row_idx[i] of Z / dZ / the dropout mask) -- a read-out that takes a few rows of the layer's output
  const uint32_t *row_idx;
  const uint32_t *dout_map;      // backward, optional [n]: row i's output gradient is dout[dout_map[i], :] (a pooled read-out's gradient table)
  // ... and dZ / dz0_amax are compact as well ([n, F] in the order of row_idx) instead of scattered into full-height buffers
  int dz_compact;
  // backward, optional (vector kernel): t_out[row, F / seg] = sum over each segment of dZ[t_branch] * Z[t_branch] -- the GAT
  // attention backward's t_i = dN_i . N_i per head (gat.hip) while both rows are in registers; t_branch < 0: none
  float *t_out; int t_branch;
};

// keep-mask (bit k: component k of the float4 at column f) of the fused output dropout
__device__ __forceinline__ uint32_t drop_keep4(const ActNormParams &p, uint64_t r, uint32_t f) {
  return drop_keep4_raw(p.seed_lo, p.seed_hi, p.drop_thr, r, f);
}

// sum over the lanes of one segment group (LS lanes, power of two)
template <int LS>
__device__ __forceinline__ float seg_sum(float v) {
#ifdef SHADOW_SEG_SUM_SHFL
#pragma unroll
  for (int off = LS / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
#else
  return group_sum<LS>(v);            // DPP butterflies (actnorm_common.h)
#endif
}

// One row per LPR lanes, one float4 per lane; LS = lanes per normalisation segment.
// (F/4 == number of active lanes per row <= LPR; seg/4 == LS)
template <int LPR, int LS, bool BWD, int NB>
__global__ void __launch_bounds__(kBlock, (BWD && NB == 2) ? 5 : 1) act_norm_kernel(ActNormParams p) {
  const uint32_t rows_per_block = kBlock / LPR;
  const uint32_t sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t f = l * 4;
  const bool lane_on = f < p.F;
  const float inv_seg = 1.0f / (float)p.seg;
  // per-thread partial sums of dscale / doffset / dbias over its rows (doffset = sum of dy is the same for
  // every branch: one accumulator)
  float4 gs[2], go, gb[2];
  gs[0] = gs[1] = go = gb[0] = gb[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  // register double buffering: the loads of this thread's NEXT row are issued before the
  // reductions of the current one, so two rows per wavefront are in flight
  const uint64_t rstep = (uint64_t)gridDim.x * rows_per_block;
  uint64_t r = (uint64_t)blockIdx.x * rows_per_block + sub;
  // (backward over selected rows: r counts the rows of the compact gradient, RR(r) is the row of Z / dZ / the mask)
  auto RR = [&](uint64_t i) -> uint64_t { return (BWD && p.row_idx) ? (uint64_t)p.row_idx[i] : i; };
  auto DR = [&](uint64_t i) -> uint64_t { return (BWD && p.dout_map) ? (uint64_t)p.dout_map[i] : i; };
  float4 zn[NB], dyn = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int b = 0; b < NB; b++) zn[b] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (backward only: the forward kernel runs at full occupancy and measured slower with it)
  constexpr bool kPrefetch = BWD;
  // (a gradient table's row map is read one row further ahead than the rows themselves: no second round trip in front of dyn)
  uint32_t mnext = (BWD && p.dout_map && r + rstep < p.n) ? p.dout_map[r + rstep] : 0u;
  if (kPrefetch && r < p.n && lane_on) {
    if (BWD && p.dout) dyn = ld4s(p.dout + (int64_t)DR(r) * p.lddo + f);
#pragma unroll
    for (int b = 0; b < NB; b++) zn[b] = ld4s(p.Z[b] + (int64_t)RR(r) * p.ldz[b] + f);
  }
  for (; r < p.n; r += rstep) {
    const uint64_t rr = RR(r);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dy = dyn;
    float4 zc[NB];
    if (kPrefetch) {
#pragma unroll
      for (int b = 0; b < NB; b++) zc[b] = zn[b];
      if (r + rstep < p.n && lane_on) {
        if (BWD && p.dout) dyn = ld4s(p.dout + (int64_t)(p.dout_map ? (uint64_t)mnext : r + rstep) * p.lddo + f);
#pragma unroll
        for (int b = 0; b < NB; b++) zn[b] = ld4s(p.Z[b] + (int64_t)RR(r + rstep) * p.ldz[b] + f);
      }
      if (BWD && p.dout_map && r + 2 * rstep < p.n) mnext = p.dout_map[r + 2 * rstep];
    } else {
#pragma unroll
      for (int b = 0; b < NB; b++) zc[b] = lane_on ? ld4s(p.Z[b] + (int64_t)rr * p.ldz[b] + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (BWD) {
      float4 ds = make_float4(p.out_scale, p.out_scale, p.out_scale, p.out_scale);
      float4 dm = ds;                      // factors of the gradient that came through the dropout mask
      if (p.drop_thr) {            // gradient of the fused output dropout: same mask, same 1/(1-p)
        const uint32_t keep = drop_keep4(p, rr, f);
        const float ks = p.out_scale * p.drop_scale;
        dm = make_float4((keep & 1u) ? ks : 0.f, (keep & 2u) ? ks : 0.f, (keep & 4u) ? ks : 0.f, (keep & 8u) ? ks : 0.f);
      }
      if (p.dout2) {               // dual mode: plain gradient (if any) + masked gradient
        float4 d2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_on) d2 = ld4s(p.dout2 + (int64_t)r * p.lddo2 + f);
        dy.x = dy.x * ds.x + d2.x * dm.x; dy.y = dy.y * ds.y + d2.y * dm.y;
        dy.z = dy.z * ds.z + d2.z * dm.z; dy.w = dy.w * ds.w + d2.w * dm.w;
      } else {
        dy.x *= dm.x; dy.y *= dm.y; dy.z *= dm.z; dy.w *= dm.w;
      }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f), h = z;
      if (lane_on) {
        z = zc[b];
        if (p.bias[b]) { const float4 bb = ld4(p.bias[b] + f); z.x += bb.x; z.y += bb.y; z.z += bb.z; z.w += bb.w; }
        h = make_float4(act_fwd(p.act[b], z.x), act_fwd(p.act[b], z.y), act_fwd(p.act[b], z.z), act_fwd(p.act[b], z.w));
      }
      // biased mean / variance over the segment (layers.py:334-335)
      const float mean = seg_sum<LS>(h.x + h.y + h.z + h.w) * inv_seg;
      float4 d = make_float4(h.x - mean, h.y - mean, h.z - mean, h.w - mean);
      if (!lane_on) d = make_float4(0.f, 0.f, 0.f, 0.f);
      const float var = seg_sum<LS>(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * inv_seg + p.eps;
      const float rstd = rsqrtf(var);
      const float4 xh = make_float4(d.x * rstd, d.y * rstd, d.z * rstd, d.w * rstd);
      float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), of = sc;
      if (lane_on) { sc = ld4(p.scale + (size_t)b * p.F + f); of = ld4(p.offset + (size_t)b * p.F + f); }
      if (!BWD) {
        // (x - mean) * scale * rsqrt(var) + offset   (layers.py:336)
        acc.x += d.x * sc.x * rstd + of.x; acc.y += d.y * sc.y * rstd + of.y;
        acc.z += d.z * sc.z * rstd + of.z; acc.w += d.w * sc.w * rstd + of.w;
      } else {
        gs[b].x += dy.x * xh.x; gs[b].y += dy.y * xh.y; gs[b].z += dy.z * xh.z; gs[b].w += dy.w * xh.w;
        if (b == 0) { go.x += dy.x; go.y += dy.y; go.z += dy.z; go.w += dy.w; }
        const float4 dxh = make_float4(dy.x * sc.x, dy.y * sc.y, dy.z * sc.z, dy.w * sc.w);
        const float m1 = seg_sum<LS>(dxh.x + dxh.y + dxh.z + dxh.w) * inv_seg;
        const float m2 = seg_sum<LS>(dxh.x * xh.x + dxh.y * xh.y + dxh.z * xh.z + dxh.w * xh.w) * inv_seg;
        float4 dh = make_float4(rstd * (dxh.x - m1 - xh.x * m2), rstd * (dxh.y - m1 - xh.y * m2),
                                rstd * (dxh.z - m1 - xh.z * m2), rstd * (dxh.w - m1 - xh.w * m2));
        float zmax = 0.f;
        if (lane_on && (p.dZ[b] || p.dbias)) {
          dh.x *= act_bwd(p.act[b], z.x, h.x); dh.y *= act_bwd(p.act[b], z.y, h.y);
          dh.z *= act_bwd(p.act[b], z.z, h.z); dh.w *= act_bwd(p.act[b], z.w, h.w);
          if (p.dZ[b]) st4s(p.dZ[b] + (int64_t)(p.dz_compact ? r : rr) * p.lddz[b] + f, dh);
          gb[b].x += dh.x; gb[b].y += dh.y; gb[b].z += dh.z; gb[b].w += dh.w;
          zmax = amax4(dh);
        }
        if (p.t_out && b == p.t_branch) {
          // t = sum over the segment of dZ_b * Z_b for an IDENTITY branch (the GAT aggregate): the normalisation does not change
          // when its input row is scaled, so its gradient is orthogonal to the row up to eps --
          //   sum_k dh_k z_k = m2 (S - sum_k xh_k^2) = S eps rstd^2 m2          (sum_k dh_k = 0, sum_k xh_k = 0, var rstd^2 = 1 - eps rstd^2)
          // -- the closed form instead of a fifth butterfly (summed directly in fp32 the same quantity is this plus rounding noise
          // of 1e-7 |dh| |z|; the plain backward kernel sits exactly on its register cap, the direct sum spilled nine dwords: + 28 %)
          if (lane_on && (l % LS) == 0) p.t_out[(p.dz_compact ? r : rr) * (p.F / p.seg) + f / p.seg] = (float)p.seg * p.eps * rstd * rstd * m2;
        }
        if (b == p.amax_branch && p.dz0_amax) {        // (the LPR lanes of a row group share r: the reduction is uniform over the group)
          zmax = group_max<LPR>(zmax);
          if (l == 0) p.dz0_amax[p.dz_compact ? r : rr] = zmax;
        }
      }
    }
    float omax = 0.f;
    if (!BWD && lane_on) {
      acc.x *= p.out_scale; acc.y *= p.out_scale; acc.z *= p.out_scale; acc.w *= p.out_scale;
      omax = amax4(acc);
      if (p.drop_thr) {
        const uint32_t keep = drop_keep4(p, r, f);
        const float4 dr = make_float4((keep & 1u) ? acc.x * p.drop_scale : 0.f, (keep & 2u) ? acc.y * p.drop_scale : 0.f,
                                      (keep & 4u) ? acc.z * p.drop_scale : 0.f, (keep & 8u) ? acc.w * p.drop_scale : 0.f);
        if (p.out2) st4s(p.out2 + (int64_t)r * p.ldo2 + f, dr);      // dual mode: out stays un-dropped
        else acc = dr;
        omax = amax4(dr);
      }
      st4s(p.out + (int64_t)r * p.ldo + f, acc);
    }
    if (!BWD && p.out_amax) {             // (the LPR lanes of a row group share r)
      omax = group_max<LPR>(omax);
      if (l == 0) p.out_amax[r] = omax;
    }
  }
  if (BWD) {
    // block reduction of the parameter gradients over the row sub-groups, then one atomic per feature
    __shared__ float red[NB][3][kBlock * 4];
#pragma unroll
    for (int b = 0; b < NB; b++) {
      float *rs = &red[b][0][threadIdx.x * 4], *ro = &red[b][1][threadIdx.x * 4], *rb = &red[b][2][threadIdx.x * 4];
      rs[0] = gs[b].x; rs[1] = gs[b].y; rs[2] = gs[b].z; rs[3] = gs[b].w;
      ro[0] = go.x; ro[1] = go.y; ro[2] = go.z; ro[3] = go.w;
      rb[0] = gb[b].x; rb[1] = gb[b].y; rb[2] = gb[b].z; rb[3] = gb[b].w;
    }
    __syncthreads();
    if (sub == 0 && lane_on) {
#pragma unroll
      for (int b = 0; b < NB; b++) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f}, o4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t q = 0; q < rows_per_block; q++) {
          const float *rs = &red[b][0][(q * LPR + l) * 4], *ro = &red[b][1][(q * LPR + l) * 4], *rb = &red[b][2][(q * LPR + l) * 4];
#pragma unroll
          for (int k = 0; k < 4; k++) { s4[k] += rs[k]; o4[k] += ro[k]; b4[k] += rb[k]; }
        }
        // deterministic two-stage reduction: this block's partial row, summed by act_norm_finish_kernel
        float *ps = p.partial + (((size_t)blockIdx.x * NB + b) * 3 + 0) * p.F + f;
        float *po = p.partial + (((size_t)blockIdx.x * NB + b) * 3 + 1) * p.F + f;
        float *pb = p.partial + (((size_t)blockIdx.x * NB + b) * 3 + 2) * p.F + f;
        st4(ps, make_float4(s4[0], s4[1], s4[2], s4[3]));
        st4(po, make_float4(o4[0], o4[1], o4[2], o4[3]));
        st4(pb, make_float4(b4[0], b4[1], b4[2], b4[3]));
      }
    }
  }
}

// dscale[b,f] = sum over blocks of partial[blk,b,0,f]; doffset (kind 1) and dbias (kind 2)
// likewise, in a fixed order.  grid (ceil(F/64), nb*3), 1024 threads = 16 groups x 64
// features: each group sums a strided subset of the partial rows, LDS combines the groups.
__global__ void act_norm_finish_kernel(const float *__restrict__ partial, uint32_t nblocks, int nb, uint32_t F,
                                       float *__restrict__ dscale, float *__restrict__ doffset,
                                       float *__restrict__ dbias) {
  __shared__ float red[16][64];
  const uint32_t fl = threadIdx.x & 63u, g = threadIdx.x >> 6;
  const uint32_t f = blockIdx.x * 64 + fl;
  const uint32_t b = blockIdx.y / 3, kind = blockIdx.y % 3;
  float *dst = kind == 0 ? dscale : (kind == 1 ? doffset : dbias);
  if (!dst) return;
  // eight running sums per thread: the partial rows are far apart (L2 round trips) -- one dependent chain of ~80 loads per
  // thread made this kernel 20 us long, four chains 13.6 us at 2 260 partial rows (five launches per products step)
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (f < F) {
    uint32_t k = g;
    for (; k + 112 < nblocks; k += 128) {
#pragma unroll
      for (int u = 0; u < 8; u++) a8[u] += partial[(((size_t)(k + 16 * u) * nb + b) * 3 + kind) * F + f];
    }
    for (int u = 0; k < nblocks; k += 16, u++) a8[u] += partial[(((size_t)k * nb + b) * 3 + kind) * F + f];
  }
  red[g][fl] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  __syncthreads();
  if (g == 0 && f < F) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) s += red[q][fl];
    dst[(size_t)b * F + f] = s;
  }
}

// generic fallback: one wavefront per row, any F / seg (seg divides F), scalar accesses.
// kLds (backward): the scale / offset / bias gradients are accumulated per wavefront in LDS ([wave][nb][3][F], every
// column belongs to one lane, rows go in order), combined per block in wavefront order and left in p.partial for
// act_norm_finish_kernel -- the same fixed-order reduction as the vector kernels, bit-reproducible.  Without it
// (F too wide for the LDS) the sums use float atomics.
template <bool BWD, bool kLds>
__global__ void act_norm_generic_kernel(ActNormParams p) {
  extern __shared__ float an_sm[];
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = lane_id();
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const float inv_seg = 1.0f / (float)p.seg;
  const uint32_t region = (uint32_t)p.nb * 3u * p.F;
  float *my = an_sm + (size_t)(threadIdx.x >> 6) * region;
  if (BWD && kLds) {
    for (uint32_t i = lane; i < region; i += 64) my[i] = 0.f;
  }
  for (uint64_t r = wave; r < p.n; r += nwaves) {
    for (uint32_t s0 = 0; s0 < p.F; s0 += p.seg) {
      for (int b = 0; b < p.nb; b++) {
        const float *z = p.Z[b] + (int64_t)r * p.ldz[b] + s0;
        const float *bi = p.bias[b] ? p.bias[b] + s0 : nullptr;
#define SHD_ZB(k) (z[k] + (bi ? bi[k] : 0.f))
        float sum = 0.f;
        for (uint32_t k = lane; k < p.seg; k += 64) sum += act_fwd(p.act[b], SHD_ZB(k));
        const float mean = wave_reduce_sum_f(sum) * inv_seg;
        float sq = 0.f;
        for (uint32_t k = lane; k < p.seg; k += 64) { const float d = act_fwd(p.act[b], SHD_ZB(k)) - mean; sq += d * d; }
        const float rstd = rsqrtf(wave_reduce_sum_f(sq) * inv_seg + p.eps);
        const float *sc = p.scale + (size_t)b * p.F + s0, *of = p.offset + (size_t)b * p.F + s0;
        if (!BWD) {
          float *o = p.out + (int64_t)r * p.ldo + s0;
          for (uint32_t k = lane; k < p.seg; k += 64) {
            const float y = ((act_fwd(p.act[b], SHD_ZB(k)) - mean) * sc[k] * rstd + of[k]) * p.out_scale;
            o[k] = (b == 0) ? y : o[k] + y;
          }
        } else {
          const float *dy = p.dout + (int64_t)(p.dout_map ? p.dout_map[r] : (uint32_t)r) * p.lddo + s0;
          float a1 = 0.f, a2 = 0.f;
          for (uint32_t k = lane; k < p.seg; k += 64) {
            const float xh = (act_fwd(p.act[b], SHD_ZB(k)) - mean) * rstd;
            const float g = dy[k] * p.out_scale;
            if (kLds) {
              my[((size_t)b * 3 + 0) * p.F + s0 + k] += g * xh;
              my[((size_t)b * 3 + 1) * p.F + s0 + k] += g;
            } else {
              atomicAdd(p.dscale + (size_t)b * p.F + s0 + k, g * xh);
              atomicAdd(p.doffset + (size_t)b * p.F + s0 + k, g);
            }
            a1 += g * sc[k]; a2 += g * sc[k] * xh;
          }
          const float m1 = wave_reduce_sum_f(a1) * inv_seg, m2 = wave_reduce_sum_f(a2) * inv_seg;
          if (p.dZ[b] || p.dbias) {
            float *dz = p.dZ[b] ? p.dZ[b] + (int64_t)r * p.lddz[b] + s0 : nullptr;
            for (uint32_t k = lane; k < p.seg; k += 64) {
              const float zz = SHD_ZB(k);
              const float h = act_fwd(p.act[b], zz);
              const float xh = (h - mean) * rstd;
              const float v = rstd * (dy[k] * p.out_scale * sc[k] - m1 - xh * m2) * act_bwd(p.act[b], zz, h);
              if (dz) dz[k] = v;
              if (p.dbias) {
                if (kLds) my[((size_t)b * 3 + 2) * p.F + s0 + k] += v;
                else atomicAdd(p.dbias + (size_t)b * p.F + s0 + k, v);
              }
            }
          }
#undef SHD_ZB
        }
      }
    }
  }
  if (BWD && kLds) {
    __syncthreads();
    const uint32_t nw = blockDim.x >> 6;
    for (uint32_t i = threadIdx.x; i < region; i += blockDim.x) {
      float acc = 0.f;
      for (uint32_t w = 0; w < nw; w++) acc += an_sm[(size_t)w * region + i];
      p.partial[(size_t)blockIdx.x * region + i] = acc;
    }
  }
}

static inline uint32_t grid_for(uint64_t work_items, uint32_t per_block, uint32_t cap = 256 * 16) {
  uint64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (uint32_t)g;
}

}  // namespace shadow

using namespace shadow;

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" int sl_gather_rows_f32(const float *d_table, int64_t ld_table, const uint32_t *d_idx,
                                  uint32_t n, uint32_t F, float *d_out, int64_t ld_out, void *stream_) {
  if (n == 0 || F == 0) return SG_OK;             // (empty tensors may come with null pointers)
  if (!d_table || !d_idx || !d_out) return set_error(SG_ERR_INVALID, "sl_gather_rows_f32: null argument");
  hipStream_t st = (hipStream_t)stream_;
  const bool vec = (F % 4 == 0) && (ld_table % 4 == 0) && (ld_out % 4 == 0) && aligned16(d_table) && aligned16(d_out);
  if (!vec) {
    hipLaunchKernelGGL(gather_rows_scalar_kernel, dim3(grid_for((uint64_t)n * F, kBlock)), dim3(kBlock), 0, st,
                       d_table, ld_table, d_idx, d_out, ld_out, n, F);
  } else if (F <= 64) {
    hipLaunchKernelGGL(gather_rows_kernel<16>, dim3(grid_for(n, kBlock / 16)), dim3(kBlock), 0, st, d_table, ld_table, d_idx, d_out, ld_out, n, F);
  } else if (F <= 128) {
    hipLaunchKernelGGL(gather_rows_kernel<32>, dim3(grid_for(n, kBlock / 32)), dim3(kBlock), 0, st, d_table, ld_table, d_idx, d_out, ld_out, n, F);
  } else {
    hipLaunchKernelGGL(gather_rows_kernel<64>, dim3(grid_for(n, kBlock / 64)), dim3(kBlock), 0, st, d_table, ld_table, d_idx, d_out, ld_out, n, F);
  }
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

// Several row copies under ONE index in one launch (sl_rows_multi): a wavefront per row walks the jobs.  What a row-sparse backward
// pass does before its kernels run -- seven saved tensors gathered on the rows' input set, two gradients scattered into zeroed
// tensors over it -- was fifteen torch launches of ~5 us each around ~1 us of copying.
struct RowsJobs { sl_rows_job j[SL_ROWS_MAX_JOBS]; int n; };

__global__ void __launch_bounds__(kBlock) rows_multi_kernel(RowsJobs J, const int64_t *__restrict__ idx, uint32_t rows) {
  const uint32_t lane = lane_id();
  for (uint32_t i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); i < rows; i += gridDim.x * (kBlock / 64)) {   // (grid_for caps the grid)
  const int64_t k = idx ? idx[i] : (int64_t)i;
  for (int q = 0; q < J.n; q++) {
    const sl_rows_job jb = J.j[q];
    const float *s = jb.mode == 0 ? jb.src + k * jb.lds : (jb.mode == 2 ? jb.src + (int64_t)i * jb.lds : nullptr);
    float *d = jb.mode == 2 ? jb.dst + k * jb.ldd : jb.dst + (int64_t)i * jb.ldd;
    const bool vec = !(jb.width & 3u) && !(jb.lds & 3) && !(jb.ldd & 3) && !((uintptr_t)jb.dst & 15u) && (jb.mode == 1 || !((uintptr_t)jb.src & 15u));
    if (vec) {
      for (uint32_t c = lane * 4; c < jb.width; c += 256)
        *reinterpret_cast<float4 *>(d + c) = s ? *reinterpret_cast<const float4 *>(s + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (uint32_t c = lane; c < jb.width; c += 64) d[c] = s ? s[c] : 0.f;
    }
  }
  }
}

extern "C" int sl_rows_multi(const sl_rows_job *jobs, int njobs, const int64_t *d_idx, uint32_t rows, void *stream_) {
  if (njobs < 0 || njobs > SL_ROWS_MAX_JOBS || (njobs && !jobs)) return set_error(SG_ERR_INVALID, "sl_rows_multi: %d jobs (at most %d)", njobs, SL_ROWS_MAX_JOBS);
  if (rows == 0 || njobs == 0) return SG_OK;
  RowsJobs J;
  J.n = njobs;
  for (int q = 0; q < njobs; q++) {
    J.j[q] = jobs[q];
    if (jobs[q].mode < 0 || jobs[q].mode > 2 || !jobs[q].dst || (jobs[q].mode != 1 && !jobs[q].src))
      return set_error(SG_ERR_INVALID, "sl_rows_multi: job %d (mode %d) incomplete", q, jobs[q].mode);
    if (jobs[q].mode != 1 && !d_idx) return set_error(SG_ERR_INVALID, "sl_rows_multi: job %d needs the row index", q);
  }
  hipLaunchKernelGGL(rows_multi_kernel, dim3(grid_for(rows, kBlock / 64)), dim3(kBlock), 0, (hipStream_t)stream_, J, d_idx, rows);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_csr_edge_rows(const uint32_t *d_indptr, uint32_t n, uint32_t e, uint32_t *d_edge_row,
                                void *stream_) {
  if (!d_indptr || (e && !d_edge_row)) return set_error(SG_ERR_INVALID, "sl_csr_edge_rows: null argument");
  if (e == 0) return SG_OK;
  hipLaunchKernelGGL(edge_rows_kernel, dim3(grid_for(e, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_, d_indptr, n, e, d_edge_row);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_csr_transpose(const uint32_t *d_indptr, const uint32_t *d_indices,
                                const uint32_t *d_edge_row, uint32_t n, uint32_t e,
                                uint32_t *d_t_indptr, uint32_t *d_t_indices, uint32_t *d_t_perm,
                                uint32_t *d_work, void *stream_) {
  if (!d_indptr || !d_t_indptr || !d_work || (e && (!d_indices || !d_edge_row || !d_t_indices || !d_t_perm)))
    return set_error(SG_ERR_INVALID, "sl_csr_transpose: null argument");
  hipStream_t st = (hipStream_t)stream_;
  // work: [0] flag, [4 .. 4+n+1) cursors, then 2*e sort scratch
  uint32_t *flag = d_work, *cursor = d_work + 4, *tmp_idx = cursor + (size_t)n + 4, *tmp_perm = tmp_idx + e;
  SHD_HIP(hipMemsetAsync(flag, 0, 16, st));
  if (e) { const int frc = fill_words(d_t_perm, 0xFFFFFFFFu, (size_t)e, st); if (frc != SG_OK) return frc; }
  const uint32_t ge = grid_for(std::max<uint64_t>(e, (uint64_t)n + 1), kBlock);
  hipLaunchKernelGGL(tr_sym_kernel, dim3(ge), dim3(kBlock), 0, st, d_indptr, d_indices, d_edge_row, n, e,
                     d_t_indptr, d_t_indices, d_t_perm, flag);
  if (e) {
    hipLaunchKernelGGL(tr_check_kernel, dim3(ge), dim3(kBlock), 0, st, d_t_perm, e, flag);
    // general path, skipped on the device when the structure was symmetric
    hipLaunchKernelGGL(tr_zero_kernel, dim3(grid_for((uint64_t)n + 1, kBlock)), dim3(kBlock), 0, st, d_t_indptr, n + 1, flag);
    hipLaunchKernelGGL(tr_count_kernel, dim3(ge), dim3(kBlock), 0, st, d_indices, n, e, d_t_indptr, flag);
    hipLaunchKernelGGL(tr_scan_kernel, dim3(1), dim3(1024), 0, st, d_t_indptr, n + 1, cursor, flag);
    hipLaunchKernelGGL(tr_fill_kernel, dim3(ge), dim3(kBlock), 0, st, d_indices, d_edge_row, e, cursor, d_t_indices, d_t_perm, flag);
    hipLaunchKernelGGL(tr_sort_rows_kernel, dim3(grid_for((uint64_t)n * 64, kBlock)), dim3(kBlock), 0, st, d_t_indptr, n,
                       d_t_indices, d_t_perm, tmp_idx, tmp_perm, flag);
    hipLaunchKernelGGL(tr_copy_back_kernel, dim3(ge), dim3(kBlock), 0, st, d_t_indptr, n, e, d_t_indices, d_t_perm, tmp_idx, tmp_perm, flag);
  }
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_degree_scales(const uint32_t *d_indptr, const float *d_edge_w, uint32_t n, int mode,
                                float *d_row_scale, void *stream_) {
  if (mode < 0 || mode > 1) return set_error(SG_ERR_INVALID, "sl_degree_scales: mode %d (0 rw, 1 sym)", mode);
  if (n == 0) return SG_OK;
  if (!d_indptr || !d_row_scale) return set_error(SG_ERR_INVALID, "sl_degree_scales: null argument");
  hipLaunchKernelGGL(degree_scales_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_,
                     d_indptr, d_edge_w, n, mode, d_row_scale);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_spmm_csr_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                               const uint32_t *d_edge_perm, const float *d_row_scale,
                               const float *d_col_scale, const float *d_X, int64_t ldx, float *d_Y,
                               int64_t ldy, uint32_t n, uint32_t F, void *stream_) {
  return sl_spmm_csr_amax_f32(d_indptr, d_indices, d_edge_w, d_edge_perm, d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, n, F, nullptr, stream_);
}

// ... with the row maxima of Y (d_row_amax [n], may be NULL) written by the same pass where the kernel holds whole rows
// (128 < F <= 256: a row per wavefront), by one more pass over Y otherwise.
extern "C" int sl_spmm_csr_amax_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                                    const uint32_t *d_edge_perm, const float *d_row_scale,
                                    const float *d_col_scale, const float *d_X, int64_t ldx, float *d_Y,
                                    int64_t ldy, uint32_t n, uint32_t F, float *d_row_amax, void *stream_) {
  return shadow::spmm_csr_amax(d_indptr, d_indices, d_edge_w, d_edge_perm, d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, n, F, d_row_amax, 0,
                               stream_);
}

// (library-internal: amax_join = 1 joins the row maxima with what d_row_amax holds -- the second half of a K-concatenated operand;
//  only where the kernel holds whole rows, spmm_csr_whole_rows)
bool shadow::spmm_csr_whole_rows(uint32_t F, const float *X, int64_t ldx, const float *Y, int64_t ldy) {
  return F > 128 && F <= 256 && (F % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(X) && aligned16(Y);
}

int shadow::spmm_csr_amax(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w, const uint32_t *d_edge_perm,
                          const float *d_row_scale, const float *d_col_scale, const float *d_X, int64_t ldx, float *d_Y, int64_t ldy,
                          uint32_t n, uint32_t F, float *d_row_amax, int amax_join, void *stream_) {
  if (n == 0 || F == 0) return SG_OK;
  if (!d_indptr || !d_X || !d_Y) return set_error(SG_ERR_INVALID, "sl_spmm_csr_f32: null argument");
  hipStream_t st = (hipStream_t)stream_;
  bool amax_done = false;
  const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(d_X) && aligned16(d_Y) && F <= 1024;
#define SHD_SPMM(LPR, CH)                                                                          \
  hipLaunchKernelGGL((spmm_kernel<LPR, CH>), dim3((((n + (kBlock / LPR) - 1) / (kBlock / LPR)) + 7u) & ~7u), dim3(kBlock), 0, st, \
                     d_indptr, d_indices, d_edge_w, d_edge_perm, d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, n, F)
  if (!vec) {
    hipLaunchKernelGGL(spmm_scalar_kernel, dim3(grid_for((uint64_t)n * F, kBlock)), dim3(kBlock), 0, st, d_indptr,
                       d_indices, d_edge_w, d_edge_perm, d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, n, F);
  } else if (F <= 32) { SHD_SPMM(8, 1); }
  else if (F <= 64) { SHD_SPMM(16, 1); }
  else if (F <= 128) { SHD_SPMM(32, 1); }       // (measured faster than half-wave units of the pipelined kernel)
  else if (F <= 256) {
    // persistent pipelined kernel: ~4 blocks (16 waves) per CU, contiguous chunks of row groups
    constexpr int R = 4, KMAX = 16;
    int ncu = 256, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t upb = kBlock / 64;                                          // units (wavefronts) per block
    const uint64_t G = ((uint64_t)n + R - 1) / R;
    const uint64_t max_units = (uint64_t)ncu * 4 * upb;                        // 4 resident blocks per CU (VGPR-bound)
    const uint32_t chunk = (uint32_t)std::max<uint64_t>(6, (G + max_units - 1) / max_units);
    const uint64_t units = (G + chunk - 1) / chunk;
    const uint32_t blocks = (uint32_t)((((units + upb - 1) / upb) + 7u) & ~(uint64_t)7u);
    hipLaunchKernelGGL((spmm_pipe_kernel<R, KMAX, 64>), dim3(blocks), dim3(kBlock), 0, st, d_indptr, d_indices, d_edge_w,
                       d_edge_perm, d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, n, F, chunk, d_row_amax, amax_join ? 1u : 0u);
    amax_done = true;
  }
  else if (F <= 512) { SHD_SPMM(64, 2); }
  else { SHD_SPMM(64, 4); }
#undef SHD_SPMM
  SHD_HIP(hipGetLastError());
  if (d_row_amax && !amax_done) {
    if (amax_join) return set_error(SG_ERR_INVALID, "spmm_csr_amax: joined row maxima need 128 < F <= 256 (F = %u)", F);
    return sl_row_amax(d_Y, ldy, n, F, d_row_amax, stream_);
  }
  return SG_OK;
}

static int spmm_blockdiag_launch(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                                 const uint32_t *d_edge_perm, const float *d_row_scale, const float *d_col_scale,
                                 const float *d_X, int64_t ldx, float *d_Y, int64_t ldy, uint32_t n, uint32_t F,
                                 const uint32_t *d_subg_node_off, const uint32_t *d_subg_edge_off, uint32_t num_subg,
                                 uint32_t max_subg_nodes, const BdGather &bg, float *d_row_amax, void *stream_, bool zero_pad = false);

static int make_drop(BdGather *bg, float drop_p, uint64_t drop_seed, const char *who) {
  memset(bg, 0, sizeof(*bg));
  bg->zero_id = 0xFFFFFFFFu;
  if (!(drop_p >= 0.f && drop_p < 1.f)) return set_error(SG_ERR_INVALID, "%s: drop_p = %g", who, drop_p);
  bg->drop_scale = 1.0f; bg->seed_lo = (uint32_t)drop_seed; bg->seed_hi = (uint32_t)(drop_seed >> 32);
  if (drop_p > 0.f) {
    bg->drop_thr = drop_threshold16(drop_p);
    bg->drop_scale = 1.0f / (1.0f - drop_p);
  }
  return SG_OK;
}

extern "C" int sl_gather_rows_drop_f32(const float *d_table, int64_t ld_table, const uint32_t *d_idx, uint32_t n, uint32_t F,
                                       float drop_p, uint64_t drop_seed, float *d_out, int64_t ld_out, uint32_t F_pad,
                                       float *d_row_amax, void *stream_) {
  if (n == 0 || F == 0) return SG_OK;
  if (!d_table || !d_idx || !d_out) return set_error(SG_ERR_INVALID, "sl_gather_rows_drop_f32: null argument");
  if ((F % 4) || (F_pad % 4) || F_pad < F || (ld_table % 4) || (ld_out % 4) || ld_out < (int64_t)F_pad || !aligned16(d_table) || !aligned16(d_out))
    return set_error(SG_ERR_INVALID, "sl_gather_rows_drop_f32: needs F %% 4 == 0, F <= F_pad <= ld_out and 16-byte aligned rows");
  BdGather bg;
  int rc;
  if ((rc = make_drop(&bg, drop_p, drop_seed, "sl_gather_rows_drop_f32")) != SG_OK) return rc;
  hipStream_t st = (hipStream_t)stream_;
  if (F_pad <= 64)
    hipLaunchKernelGGL(gather_rows_drop_kernel<16>, dim3(grid_for(n, kBlock / 16)), dim3(kBlock), 0, st, d_table, ld_table, d_idx, d_out, ld_out, n, F, F_pad, bg, d_row_amax);
  else if (F_pad <= 128)
    hipLaunchKernelGGL(gather_rows_drop_kernel<32>, dim3(grid_for(n, kBlock / 32)), dim3(kBlock), 0, st, d_table, ld_table, d_idx, d_out, ld_out, n, F, F_pad, bg, d_row_amax);
  else
    hipLaunchKernelGGL(gather_rows_drop_kernel<64>, dim3(grid_for(n, kBlock / 64)), dim3(kBlock), 0, st, d_table, ld_table, d_idx, d_out, ld_out, n, F, F_pad, bg, d_row_amax);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

// Consecutive small subgraphs joined into groups of at most cap_rows rows and cap_edges edges: the ranges of a group are
// still a diagonal block (its rows reference its own columns only), so the block-diagonal SpMM can stage a whole group in
// LDS at once -- PPR subgraphs of ~150 rows fill barely more than one 128-row pass of the kernel each, pairs fill 2.3 of 3.
// Groups are ALIGNED power-of-two runs of subgraphs: subgraph s belongs to the largest run [s & ~(2^r - 1), + 2^r) that fits
// the caps (runs nest, so every member of a run finds the same one) -- one thread per subgraph, no serial walk.
// Output: [P + 1] offsets each (groups first, then empty groups up to P: the launch needs no count from the device).
__global__ void __launch_bounds__(1024) merge_subgraphs_kernel(const uint32_t *__restrict__ node_off, const uint32_t *__restrict__ edge_off,
                                                               uint32_t P, uint32_t cap_rows, uint32_t cap_edges,
                                                               uint32_t *__restrict__ gnode, uint32_t *__restrict__ gedge) {
  extern __shared__ uint32_t ms[];                       // [2][P + 1] offsets, then [1024] scan cells
  uint32_t *no = ms, *eo = ms + (P + 1), *cell = eo + (P + 1);
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i <= P; i += 1024) { no[i] = node_off[i]; eo[i] = edge_off[i]; }
  __syncthreads();
  // thread t owns the subgraphs [8 t, 8 t + 8): group starts among them, exclusive scan of the counts over the workgroup
  uint32_t starts = 0, cnt = 0;
#pragma unroll
  for (uint32_t q = 0; q < 8; q++) {
    const uint32_t s = 8 * tid + q;
    if (s < P) {
      uint32_t r = 0;
      for (;;) {
        const uint32_t a = s & ~((2u << r) - 1u), b = a + (2u << r);          // the next larger aligned run
        if (b > P || no[b] - no[a] > cap_rows || eo[b] - eo[a] > cap_edges) break;
        r++;
      }
      if ((s & ((1u << r) - 1u)) == 0) { starts |= 1u << q; cnt++; }
    }
  }
  cell[tid] = cnt;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t v = tid >= off ? cell[tid - off] : 0u;
    __syncthreads();
    cell[tid] += v;
    __syncthreads();
  }
  uint32_t g = cell[tid] - cnt;
  const uint32_t ngroups = cell[1023];
#pragma unroll
  for (uint32_t q = 0; q < 8; q++)
    if (starts & (1u << q)) { gnode[g] = no[8 * tid + q]; gedge[g] = eo[8 * tid + q]; g++; }
  for (uint32_t i = ngroups + tid; i <= P; i += 1024) { gnode[i] = no[P]; gedge[i] = eo[P]; }
}

extern "C" int sl_merge_subgraphs(const uint32_t *d_node_off, const uint32_t *d_edge_off, uint32_t num_subg, uint32_t cap_rows,
                                  uint32_t cap_edges, uint32_t *d_group_node_off, uint32_t *d_group_edge_off, void *stream_) {
  if (!d_node_off || !d_edge_off || !d_group_node_off || !d_group_edge_off) return set_error(SG_ERR_INVALID, "sl_merge_subgraphs: null argument");
  if (num_subg == 0) return SG_OK;
  if (num_subg > 8191) return set_error(SG_ERR_INVALID, "sl_merge_subgraphs: at most 8191 subgraphs");
  if (cap_rows == 0 || cap_rows > (uint32_t)kBdMaxRows) cap_rows = kBdMaxRows;
  if (cap_edges == 0 || cap_edges > (uint32_t)kBdMaxEdges) cap_edges = kBdMaxEdges;
  const size_t merge_lds = (size_t)2 * (num_subg + 1) * 4 + 1024 * 4;          // (68 KB at the 8191-subgraph limit)
  if (merge_lds > 64 * 1024) SHD_HIP(ensure_dynamic_lds((const void *)merge_subgraphs_kernel, merge_lds));
  hipLaunchKernelGGL(merge_subgraphs_kernel, dim3(1), dim3(1024), merge_lds, (hipStream_t)stream_, d_node_off,
                     d_edge_off, num_subg, cap_rows, cap_edges, d_group_node_off, d_group_edge_off);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_spmm_blockdiag_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                                     const uint32_t *d_edge_perm, const float *d_row_scale,
                                     const float *d_col_scale, const float *d_X, int64_t ldx, float *d_Y,
                                     int64_t ldy, uint32_t n, uint32_t F, const uint32_t *d_subg_node_off,
                                     const uint32_t *d_subg_edge_off, uint32_t num_subg,
                                     uint32_t max_subg_nodes, float *d_row_amax, void *stream_) {
  if (!d_indptr || !d_X || !d_Y || !d_subg_node_off || !d_subg_edge_off)
    return set_error(SG_ERR_INVALID, "sl_spmm_blockdiag_f32: null argument");
  if (n == 0 || F == 0 || num_subg == 0) return SG_OK;
  const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(d_X) && aligned16(d_Y);
  if (!vec) {  // unaligned layouts: the general kernel handles them
    if (d_row_amax) return set_error(SG_ERR_INVALID, "sl_spmm_blockdiag_f32: row maxima need the vector layout (F, ld %% 4 == 0, 16-byte aligned)");
    return sl_spmm_csr_f32(d_indptr, d_indices, d_edge_w, d_edge_perm, d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy,
                           n, F, stream_);
  }
  BdGather bg;
  memset(&bg, 0, sizeof(bg));
  return spmm_blockdiag_launch(d_indptr, d_indices, d_edge_w, d_edge_perm, d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, n, F,
                               d_subg_node_off, d_subg_edge_off, num_subg, max_subg_nodes, bg, d_row_amax, stream_);
}

namespace shadow {
// sl_spmm_blockdiag_f32 on operands whose rows sit on whole 128-byte lines (ld % 32 == 0, 128-byte aligned, F % 32 != 0):
// additionally writes zeros into the pad of every row's last line, [F, roundup32(F)) -- for callers that own the pad
// (sl_sage_fwd: the layer's GEMM then reads A X at the padded width, without a K tail).  false: layout not eligible.
bool spmm_blockdiag_lines_ok(uint32_t F, const float *X, int64_t ldx, const float *Y, int64_t ldy) {
  const char *e = getenv("SHADOW_SPMM_LINES");
  return !(e && e[0] == '0') && (F % 4 == 0) && F % 32 != 0 && ldx % 32 == 0 && ldy % 32 == 0 && (reinterpret_cast<uintptr_t>(X) & 127) == 0 &&
         (reinterpret_cast<uintptr_t>(Y) & 127) == 0;
}
int spmm_blockdiag_padded(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w, const uint32_t *d_edge_perm,
                          const float *d_row_scale, const float *d_col_scale, const float *d_X, int64_t ldx, float *d_Y, int64_t ldy,
                          uint32_t n, uint32_t F, const uint32_t *d_subg_node_off, const uint32_t *d_subg_edge_off, uint32_t num_subg,
                          uint32_t max_subg_nodes, float *d_row_amax, void *stream_) {
  if (n == 0 || F == 0 || num_subg == 0) return SG_OK;
  BdGather bg;
  memset(&bg, 0, sizeof(bg));
  return spmm_blockdiag_launch(d_indptr, d_indices, d_edge_w, d_edge_perm, d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, n, F,
                               d_subg_node_off, d_subg_edge_off, num_subg, max_subg_nodes, bg, d_row_amax, stream_, true);
}
}  // namespace shadow

extern "C" int sl_spmm_blockdiag_gather_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                                            const uint32_t *d_edge_perm, const float *d_row_scale, const float *d_col_scale,
                                            const float *d_table, int64_t ldt, const uint32_t *d_ids, float drop_p,
                                            uint64_t drop_seed, float *d_Xout, int64_t ldxo, float *d_Y, int64_t ldy,
                                            uint32_t n, uint32_t F, const uint32_t *d_subg_node_off,
                                            const uint32_t *d_subg_edge_off, uint32_t num_subg, uint32_t max_subg_nodes,
                                            void *stream_) {
  if (!d_indptr || !d_table || !d_ids || !d_Y || !d_subg_node_off || !d_subg_edge_off)
    return set_error(SG_ERR_INVALID, "sl_spmm_blockdiag_gather_f32: null argument");
  if (n == 0 || F == 0 || num_subg == 0) return SG_OK;
  if ((F % 4) || (ldt % 4) || (ldy % 4) || !aligned16(d_table) || !aligned16(d_Y) || (d_Xout && ((ldxo % 4) || !aligned16(d_Xout))))
    return set_error(SG_ERR_INVALID, "sl_spmm_blockdiag_gather_f32: needs F %% 4 == 0 and 16-byte aligned rows");
  BdGather bg;
  int rc;
  if ((rc = make_drop(&bg, drop_p, drop_seed, "sl_spmm_blockdiag_gather_f32")) != SG_OK) return rc;
  bg.table = d_table; bg.ldt = ldt; bg.ids = d_ids; bg.xout = d_Xout; bg.ldxo = ldxo;
  return spmm_blockdiag_launch(d_indptr, d_indices, d_edge_w, d_edge_perm, d_row_scale, d_col_scale, d_table, ldt, d_Y, ldy, n, F,
                               d_subg_node_off, d_subg_edge_off, num_subg, max_subg_nodes, bg, nullptr, stream_);
}

// Y = A . table[ids] on a block-diagonal A: the input rows are taken through a row map (ids[r] = row of `table` that stands for
// batch row r; many rows may share one -- e.g. a zero row for the rows a row-sparse gradient does not reach), joined row maxima
// of Y as in sl_spmm_blockdiag_f32.
extern "C" int sl_spmm_blockdiag_rows_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                                          const uint32_t *d_edge_perm, const float *d_row_scale, const float *d_col_scale,
                                          const float *d_table, int64_t ldt, const uint32_t *d_ids, float *d_Y, int64_t ldy, uint32_t n,
                                          uint32_t F, const uint32_t *d_subg_node_off, const uint32_t *d_subg_edge_off, uint32_t num_subg,
                                          uint32_t max_subg_nodes, float *d_row_amax, uint32_t zero_id, void *stream_) {
  if (!d_indptr || !d_table || !d_ids || !d_Y || !d_subg_node_off || !d_subg_edge_off)
    return set_error(SG_ERR_INVALID, "sl_spmm_blockdiag_rows_f32: null argument");
  if (n == 0 || F == 0 || num_subg == 0) return SG_OK;
  if ((F % 4) || (ldt % 4) || (ldy % 4) || !aligned16(d_table) || !aligned16(d_Y))
    return set_error(SG_ERR_INVALID, "sl_spmm_blockdiag_rows_f32: needs F %% 4 == 0 and 16-byte aligned rows");
  BdGather bg;
  int rc;
  if ((rc = make_drop(&bg, 0.f, 0, "sl_spmm_blockdiag_rows_f32")) != SG_OK) return rc;
  bg.table = d_table; bg.ldt = ldt; bg.ids = d_ids; bg.xout = nullptr; bg.ldxo = 0; bg.zero_id = zero_id;
  return spmm_blockdiag_launch(d_indptr, d_indices, d_edge_w, d_edge_perm, d_row_scale, d_col_scale, d_table, ldt, d_Y, ldy, n, F,
                               d_subg_node_off, d_subg_edge_off, num_subg, max_subg_nodes, bg, d_row_amax, stream_);
}

static int spmm_blockdiag_launch(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                                 const uint32_t *d_edge_perm, const float *d_row_scale, const float *d_col_scale,
                                 const float *d_X, int64_t ldx, float *d_Y, int64_t ldy, uint32_t n, uint32_t F,
                                 const uint32_t *d_subg_node_off, const uint32_t *d_subg_edge_off, uint32_t num_subg,
                                 uint32_t max_subg_nodes, const BdGather &bg, float *d_row_amax, void *stream_, bool zero_pad) {
  hipStream_t st = (hipStream_t)stream_;
  int ncu = 256, dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  // rows of the LDS tile: the batch's largest subgraph, at most kBdMaxRows (bigger ones gather from HBM)
  uint32_t cap_rows = std::min<uint32_t>(std::max<uint32_t>(max_subg_nodes, 32), (uint32_t)kBdMaxRows);
  cap_rows = (cap_rows + 31u) & ~31u;
  const size_t lds = (size_t)cap_rows * kBdRowPad * 4 + ((size_t)cap_rows + 4) * 4 + (size_t)kBdMaxEdges * 8;
  if (lds > 64 * 1024) {
    SHD_HIP(ensure_dynamic_lds((const void *)spmm_blockdiag_kernel<0>, lds));
    SHD_HIP(ensure_dynamic_lds((const void *)spmm_blockdiag_kernel<1>, lds));
  }
  const uint32_t tiles = (F / 4 + 7) / 8;            // <= 8 float4 columns per tile (bd_col4)
  // both operands on whole 128-byte lines per row: tiles = lines (SHADOW_SPMM_LINES=0: the even split)
  const char *lines_env = getenv("SHADOW_SPMM_LINES");      // (read per call: a test compares the two splits in one process)
  const bool lines_on = !(lines_env && lines_env[0] == '0');
  uint32_t lines = (lines_on && !bg.table && F % 32 != 0 && ldx % 32 == 0 && ldy % 32 == 0 && (reinterpret_cast<uintptr_t>(d_X) & 127) == 0 &&
                    (reinterpret_cast<uintptr_t>(d_Y) & 127) == 0) ? 1u : 0u;
  // zero_pad (shadow::spmm_blockdiag_padded): the idle lanes of a row's last line write zeros into the pad -- whole-line
  // writes, and the consumer may read the rows at the padded width (the K-tail-free GEMM of sl_sage_fwd)
  if (zero_pad) {
    if (!lines) return set_error(SG_ERR_INVALID, "spmm_blockdiag_padded: the operands are not on whole 128-byte lines per row");
    lines = 2u;
  }
  // resident workgroups per CU: what LDS allows, at most what 2048 threads allow (a grid beyond the resident set would
  // run as a partial second round of a persistent kernel, which costs a full one)
  const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(2048 / kBdBlock, (size_t)(160 * 1024) / (lds + 256)));
  // tiles a workgroup takes back to back (the structure is staged once per group): as many as keep enough groups per
  // resident workgroup for the ragged subgraph sizes to even out -- 4 for the k-hop batches (~280 rows per subgraph),
  // 12 for small subgraphs such as PPR's (~150 rows, sizes 1 .. k) (round-2 sweep: products PPR batch
  // 0.175 ms at groups of 4, 0.153 at single tiles; k-hop batch 0.147 vs 0.180)
  uint32_t tg = tiles;
  const uint64_t want = ((uint64_t)n >= (uint64_t)224 * num_subg) ? 4 : 12;
  while (tg > 1 && (uint64_t)num_subg * ((tiles + tg - 1) / tg) < want * ncu * per_cu) tg = (tg + 1) / 2;
  const uint64_t total = (uint64_t)num_subg * ((tiles + tg - 1) / tg);
  if (total + 2 * (uint64_t)ncu * per_cu >= ((uint64_t)1 << 32))
    return set_error(SG_ERR_INVALID, "sl_spmm_blockdiag_f32: too many (subgraph, tile) items");
  const uint32_t grid = (uint32_t)std::min<uint64_t>(total, (uint64_t)ncu * per_cu);
  if (bg.table && bg.zero_id != 0xFFFFFFFFu && !bg.drop_thr && !bg.xout) {
    if (lds > 64 * 1024) SHD_HIP(ensure_dynamic_lds((const void *)spmm_blockdiag_kernel<2>, lds));
    hipLaunchKernelGGL(spmm_blockdiag_kernel<2>, dim3(grid), dim3(kBdBlock), lds, st, d_indptr, d_indices, d_edge_w, d_edge_perm,
                       d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, F, d_subg_node_off, d_subg_edge_off, num_subg, tiles,
                       tg, cap_rows, bg, reinterpret_cast<uint32_t *>(d_row_amax), lines);
  } else if (bg.table)
    hipLaunchKernelGGL(spmm_blockdiag_kernel<1>, dim3(grid), dim3(kBdBlock), lds, st, d_indptr, d_indices, d_edge_w, d_edge_perm,
                       d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, F, d_subg_node_off, d_subg_edge_off, num_subg, tiles,
                       tg, cap_rows, bg, reinterpret_cast<uint32_t *>(d_row_amax), lines);
  else
    hipLaunchKernelGGL(spmm_blockdiag_kernel<0>, dim3(grid), dim3(kBdBlock), lds, st, d_indptr, d_indices, d_edge_w, d_edge_perm,
                       d_row_scale, d_col_scale, d_X, ldx, d_Y, ldy, F, d_subg_node_off, d_subg_edge_off, num_subg, tiles,
                       tg, cap_rows, bg, reinterpret_cast<uint32_t *>(d_row_amax), lines);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

namespace shadow {
// the second stage of the parameter-gradient reduction for callers outside this file (gemm_fused.hip: the act_norm
// backward that rides in a GEMM epilogue leaves one partial row per workgroup)
int act_norm_finish_launch(const float *partial, uint32_t nblocks, int nb, uint32_t F, float *dscale, float *doffset, float *dbias,
                           hipStream_t st) {
  hipLaunchKernelGGL(act_norm_finish_kernel, dim3((F + 63) / 64, nb * 3), dim3(1024), 0, st, partial, nblocks, nb, F, dscale, doffset,
                     dbias);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}
}  // namespace shadow

constexpr uint32_t kActNormBwdBlocks = 2048;  // == rows of the partial buffer

// blocks of `kernel` (kBlock threads, no dynamic LDS) that fit on the device at once
static uint32_t resident_blocks(const void *kernel) {
  int per_cu = 0, ncu = 256, dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 4;
  return (uint32_t)per_cu * (uint32_t)ncu;
}

static int act_norm_launch(ActNormParams &p, bool bwd, hipStream_t st, bool *vector_kernel = nullptr, bool need_vector = false) {
  const uint32_t F = p.F, seg = p.seg;
  bool vec = (F % 4 == 0) && (seg % 4 == 0) && F <= 256 && (F % seg == 0);
  // lanes per segment must be a power of two; when seg == F and F/4 is not a
  // power of two the idle lanes of the row group contribute zeros
  uint32_t lpr = 1; while (lpr * 4 < F) lpr <<= 1;
  uint32_t ls = seg == F ? lpr : seg / 4;
  if (seg != F && (ls & (ls - 1))) vec = false;
  for (int b = 0; b < p.nb; b++) {
    vec = vec && aligned16(p.Z[b]) && (p.ldz[b] % 4 == 0);
    if (bwd && p.dZ[b]) vec = vec && aligned16(p.dZ[b]) && (p.lddz[b] % 4 == 0);
  }
  vec = vec && aligned16(p.scale) && aligned16(p.offset);
  for (int b = 0; b < p.nb; b++) if (p.bias[b]) vec = vec && aligned16(p.bias[b]);
  if (!bwd) vec = vec && aligned16(p.out) && (p.ldo % 4 == 0);
  else if (p.dout) vec = vec && aligned16(p.dout) && (p.lddo % 4 == 0);
  if (lpr < 4) vec = false;
  // the kernels are grid-stride loops: launch exactly the resident number of blocks (a partial second
  // round of blocks costs as much as a full one -- measured 0.19 -> 0.27 ms when one VGPR too many
  // dropped the occupancy below the old fixed grid)
#define SHD_AN_LAUNCH(KERNEL, CAP)                                                                         \
  do {                                                                                                     \
    g = grid_for(p.n, kBlock / lpr, std::min<uint32_t>(CAP, resident_blocks((const void *)KERNEL)));       \
    hipLaunchKernelGGL(KERNEL, dim3(g), dim3(kBlock), 0, st, p);                                           \
  } while (0)
#define SHD_AN(LPR, LS)                                                                                    \
  do {                                                                                                     \
    uint32_t g = 1;                                                                                        \
    if (bwd) {                                                                                             \
      if (p.nb == 1) SHD_AN_LAUNCH((act_norm_kernel<LPR, LS, true, 1>), kActNormBwdBlocks);                \
      else SHD_AN_LAUNCH((act_norm_kernel<LPR, LS, true, 2>), kActNormBwdBlocks);                          \
      hipLaunchKernelGGL(act_norm_finish_kernel, dim3((p.F + 63) / 64, p.nb * 3), dim3(1024), 0, st,       \
                         p.partial, g, p.nb, p.F, p.dscale, p.doffset, p.dbias);                           \
    } else if (p.nb == 1) SHD_AN_LAUNCH((act_norm_kernel<LPR, LS, false, 1>), 256 * 8);                    \
    else SHD_AN_LAUNCH((act_norm_kernel<LPR, LS, false, 2>), 256 * 8);                                     \
  } while (0)
  bool done = false;
  if (vec) {
    done = true;
    if (lpr == 64 && ls == 64) SHD_AN(64, 64);
    else if (lpr == 64 && ls == 32) SHD_AN(64, 32);
    else if (lpr == 64 && ls == 16) SHD_AN(64, 16);
    else if (lpr == 64 && ls == 8) SHD_AN(64, 8);
    else if (lpr == 32 && ls == 32) SHD_AN(32, 32);
    else if (lpr == 32 && ls == 16) SHD_AN(32, 16);
    else if (lpr == 32 && ls == 8) SHD_AN(32, 8);
    else if (lpr == 16 && ls == 16) SHD_AN(16, 16);
    else if (lpr == 16 && ls == 8) SHD_AN(16, 8);
    else if (lpr == 8 && ls == 8) SHD_AN(8, 8);
    else if (lpr == 4 && ls == 4) SHD_AN(4, 4);
    else done = false;
  }
#undef SHD_AN
#undef SHD_AN_LAUNCH
  if (vector_kernel) *vector_kernel = done;
  if (!done && need_vector)
    return set_error(SG_ERR_INVALID, "sl_act_norm_bwd: the selected-rows form needs the vector layout (F %% 4 == 0, F <= 256, aligned operands)");
  if (!done && p.drop_thr)
    return set_error(SG_ERR_INVALID, "sl_act_norm: fused output dropout needs the vector layout (F %% 4 == 0, F <= 256, "
                                     "16-byte aligned operands); apply dropout separately for this shape");
  if (!done) {
    const uint32_t g = grid_for((uint64_t)p.n * 64, kBlock, 256 * 8);
    const size_t lds = (size_t)(kBlock / 64) * p.nb * 3 * p.F * sizeof(float);
    if (bwd && lds <= 144 * 1024) {
      // deterministic: per-wavefront LDS partials -> per-block partials -> fixed-order finish
      if (lds > 64 * 1024)
        SHD_HIP(ensure_dynamic_lds((const void *)act_norm_generic_kernel<true, true>, lds));
      hipLaunchKernelGGL((act_norm_generic_kernel<true, true>), dim3(g), dim3(kBlock), lds, st, p);
      hipLaunchKernelGGL(act_norm_finish_kernel, dim3((p.F + 63) / 64, p.nb * 3), dim3(1024), 0, st, p.partial, g, p.nb, p.F,
                         p.dscale, p.doffset, p.dbias);
    } else if (bwd) {
      SHD_HIP(hipMemsetAsync(p.dscale, 0, (size_t)p.nb * p.F * 4, st));
      SHD_HIP(hipMemsetAsync(p.doffset, 0, (size_t)p.nb * p.F * 4, st));
      if (p.dbias) SHD_HIP(hipMemsetAsync(p.dbias, 0, (size_t)p.nb * p.F * 4, st));
      hipLaunchKernelGGL((act_norm_generic_kernel<true, false>), dim3(g), dim3(kBlock), 0, st, p);
    } else {
      hipLaunchKernelGGL((act_norm_generic_kernel<false, false>), dim3(g), dim3(kBlock), 0, st, p);
    }
  }
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

static int act_norm_check(int nb, uint32_t F, uint32_t seg, const float *const *Z, const int *act, uint32_t n) {
  if (nb < 1 || nb > 2) return set_error(SG_ERR_INVALID, "sl_act_norm: nb must be 1 or 2");
  if (seg == 0 || F % seg != 0) return set_error(SG_ERR_INVALID, "sl_act_norm: seg=%u must divide F=%u", seg, F);
  for (int b = 0; b < nb; b++) {
    if (n && !Z[b]) return set_error(SG_ERR_INVALID, "sl_act_norm: null branch input");
    if (act[b] < 0 || act[b] > 4) return set_error(SG_ERR_INVALID, "sl_act_norm: unknown activation %d", act[b]);
  }
  return SG_OK;
}

static int set_dropout(ActNormParams &p, float drop_p, uint64_t drop_seed, const char *who) {
  p.drop_thr = 0; p.drop_scale = 1.0f; p.seed_lo = (uint32_t)drop_seed; p.seed_hi = (uint32_t)(drop_seed >> 32);
  if (drop_p <= 0.f) return SG_OK;
  if (!(drop_p < 1.f)) return set_error(SG_ERR_INVALID, "%s: dropout probability %g", who, drop_p);
  p.drop_thr = drop_threshold16(drop_p);
  p.drop_scale = 1.0f / (1.0f - drop_p);
  return SG_OK;
}

// Does the vector kernel (float4 rows; fused dropout, row maxima, selected-rows backward live there) exist for this shape?
// (operand alignment is checked per call)
extern "C" int sl_act_norm_vector_layout(uint32_t F, uint32_t seg) {
  if (F == 0 || seg == 0 || (F % 4) || (seg % 4) || F > 256 || (F % seg)) return 0;
  uint32_t lpr = 1; while (lpr * 4 < F) lpr <<= 1;
  const uint32_t ls = seg == F ? lpr : seg / 4;
  if (lpr < 4 || (ls & (ls - 1))) return 0;
  return (lpr == 64 && ls >= 8) || (lpr == 32 && ls >= 8) || (lpr == 16 && ls >= 8) || (lpr == 8 && ls == 8) || (lpr == 4 && ls == 4);
}

extern "C" int sl_act_norm_fwd(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                               const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                               uint32_t seg, float out_scale, float *d_out, int64_t ldo, float drop_p,
                               uint64_t drop_seed, float *d_out_dropped, int64_t ldo_dropped, float *d_out_amax, void *stream_) {
  int rc = act_norm_check(nb, F, seg, d_Z, act, n);
  if (rc) return rc;
  if (n == 0) return SG_OK;
  if (!d_scale || !d_offset || !d_out) return set_error(SG_ERR_INVALID, "sl_act_norm_fwd: null argument");
  ActNormParams p;
  memset(&p, 0, sizeof(p));
  for (int b = 0; b < nb; b++) { p.Z[b] = d_Z[b]; p.ldz[b] = ldz[b]; p.act[b] = act[b]; p.bias[b] = d_bias ? d_bias[b] : nullptr; }
  p.scale = d_scale; p.offset = d_offset; p.nb = nb; p.n = n; p.F = F; p.seg = seg;
  p.out_scale = out_scale; p.eps = 1e-9f; p.out = d_out; p.ldo = ldo;
  if ((rc = set_dropout(p, drop_p, drop_seed, "sl_act_norm_fwd")) != SG_OK) return rc;
  if (d_out_dropped) {
    if (!p.drop_thr) return set_error(SG_ERR_INVALID, "sl_act_norm_fwd: a dropped output needs drop_p > 0");
    if ((ldo_dropped & 3) || !aligned16(d_out_dropped))
      return set_error(SG_ERR_INVALID, "sl_act_norm_fwd: the dropped output must be 16-byte aligned, ld %% 4 == 0");
    p.out2 = d_out_dropped; p.ldo2 = ldo_dropped;
  }
  p.out_amax = d_out_amax;
  bool vec = false;
  if ((rc = act_norm_launch(p, false, (hipStream_t)stream_, &vec)) != SG_OK) return rc;
  // (the general kernel does not write the row maxima: one more pass, over the tensor the next layer reads)
  if (d_out_amax && !vec) {
    const float *o = d_out_dropped ? d_out_dropped : d_out;
    return sl_row_amax(o, d_out_dropped ? ldo_dropped : ldo, n, F, d_out_amax, stream_);
  }
  return SG_OK;
}

extern "C" int sl_act_norm_bwd_rows(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                                    const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                                    uint32_t seg, float out_scale, const float *d_dout, int64_t lddo,
                                    float *const *d_dZ, const int64_t *lddz, float *d_dscale,
                                    float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                                    const float *d_dout_dropped, int64_t lddo_dropped, float *d_dz0_amax, const uint32_t *d_row_idx,
                                    int dz_compact, void *stream_);

extern "C" int sl_act_norm_bwd(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                               const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                               uint32_t seg, float out_scale, const float *d_dout, int64_t lddo,
                               float *const *d_dZ, const int64_t *lddz, float *d_dscale,
                               float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                               const float *d_dout_dropped, int64_t lddo_dropped, float *d_dz0_amax, const uint32_t *d_row_idx,
                               void *stream_) {
  return sl_act_norm_bwd_rows(nb, d_Z, ldz, d_bias, act, d_scale, d_offset, n, F, seg, out_scale, d_dout, lddo, d_dZ, lddz, d_dscale, d_doffset,
                              d_dbias, d_partial, drop_p, drop_seed, d_dout_dropped, lddo_dropped, d_dz0_amax, d_row_idx, 0, stream_);
}

extern "C" int sl_act_norm_bwd_rows(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                                    const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                                    uint32_t seg, float out_scale, const float *d_dout, int64_t lddo,
                                    float *const *d_dZ, const int64_t *lddz, float *d_dscale,
                                    float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                                    const float *d_dout_dropped, int64_t lddo_dropped, float *d_dz0_amax, const uint32_t *d_row_idx,
                                    int dz_compact, void *stream_) {
  return sl_act_norm_bwd_rows_t(nb, d_Z, ldz, d_bias, act, d_scale, d_offset, n, F, seg, out_scale, d_dout, lddo, d_dZ, lddz, d_dscale,
                                d_doffset, d_dbias, d_partial, drop_p, drop_seed, d_dout_dropped, lddo_dropped, d_dz0_amax, d_row_idx,
                                dz_compact, -1, nullptr, 0, stream_);
}

static int act_norm_bwd_impl(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                             const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                             uint32_t seg, float out_scale, const float *d_dout, int64_t lddo,
                             float *const *d_dZ, const int64_t *lddz, float *d_dscale,
                             float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                             const float *d_dout_dropped, int64_t lddo_dropped, float *d_dz0_amax, const uint32_t *d_row_idx,
                             int dz_compact, int t_branch, float *d_t_out, const uint32_t *d_dout_map, int amax_branch, void *stream_) {
  if (amax_branch < 0 || amax_branch >= nb) return set_error(SG_ERR_INVALID, "sl_act_norm_bwd: row maxima of branch %d of %d", amax_branch, nb);
  if (d_dout_map && (!d_dout || d_row_idx)) return set_error(SG_ERR_INVALID, "sl_act_norm_bwd_map: a gradient table needs d_dout and no row selection");
  if (d_t_out && (t_branch < 0 || t_branch >= nb || act[t_branch] != 0 || (d_bias && d_bias[t_branch])))
    return set_error(SG_ERR_INVALID, "sl_act_norm_bwd_rows_t: the segment dots are provided for an identity branch without a bias");
  if (dz_compact && !d_row_idx) return set_error(SG_ERR_INVALID, "sl_act_norm_bwd_rows: compact dZ without row indices");
  int rc = act_norm_check(nb, F, seg, d_Z, act, n);
  if (rc) return rc;
  if (!d_scale || !d_offset || (!d_dout && !d_dout_dropped) || !d_dscale || !d_doffset || !d_dZ)
    return set_error(SG_ERR_INVALID, "sl_act_norm_bwd: null argument");
  hipStream_t st = (hipStream_t)stream_;
  if (n == 0) {
    SHD_HIP(hipMemsetAsync(d_dscale, 0, (size_t)nb * F * 4, st));
    SHD_HIP(hipMemsetAsync(d_doffset, 0, (size_t)nb * F * 4, st));
    if (d_dbias) SHD_HIP(hipMemsetAsync(d_dbias, 0, (size_t)nb * F * 4, st));
    return SG_OK;
  }
  if (!d_partial) return set_error(SG_ERR_INVALID, "sl_act_norm_bwd: null partial buffer");
  ActNormParams p;
  memset(&p, 0, sizeof(p));
  for (int b = 0; b < nb; b++) {
    p.Z[b] = d_Z[b]; p.ldz[b] = ldz[b]; p.act[b] = act[b]; p.bias[b] = d_bias ? d_bias[b] : nullptr;
    p.dZ[b] = d_dZ[b]; p.lddz[b] = lddz ? lddz[b] : 0;
  }
  p.dbias = d_dbias;
  p.scale = d_scale; p.offset = d_offset; p.nb = nb; p.n = n; p.F = F; p.seg = seg;
  p.out_scale = out_scale; p.eps = 1e-9f; p.dout = d_dout; p.lddo = lddo;
  p.dscale = d_dscale; p.doffset = d_doffset; p.partial = d_partial;
  if ((rc = set_dropout(p, drop_p, drop_seed, "sl_act_norm_bwd")) != SG_OK) return rc;
  if (d_dout_dropped) {
    if (!p.drop_thr) return set_error(SG_ERR_INVALID, "sl_act_norm_bwd: a dropped-output gradient needs drop_p > 0");
    if ((lddo_dropped & 3) || !aligned16(d_dout_dropped))
      return set_error(SG_ERR_INVALID, "sl_act_norm_bwd: the dropped-output gradient must be 16-byte aligned, ld %% 4 == 0");
    p.dout2 = d_dout_dropped; p.lddo2 = lddo_dropped;
  }
  if (d_dz0_amax && !d_dZ[amax_branch]) return set_error(SG_ERR_INVALID, "sl_act_norm_bwd: row maxima of a gradient that is not written");
  p.dz0_amax = d_dz0_amax; p.amax_branch = amax_branch;
  p.row_idx = d_row_idx;
  p.dout_map = d_dout_map;
  p.dz_compact = dz_compact ? 1 : 0;
  p.t_out = d_t_out; p.t_branch = d_t_out ? t_branch : -1;
  bool vec = false;
  if ((rc = act_norm_launch(p, true, st, &vec, d_row_idx != nullptr || d_t_out != nullptr)) != SG_OK) return rc;
  // (the general kernel does not write the row maxima: one more pass)
  return (d_dz0_amax && !vec) ? sl_row_amax(d_dZ[amax_branch], lddz[amax_branch], n, F, d_dz0_amax, st) : SG_OK;
}

extern "C" int sl_act_norm_bwd_rows_t(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                                      const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                                      uint32_t seg, float out_scale, const float *d_dout, int64_t lddo,
                                      float *const *d_dZ, const int64_t *lddz, float *d_dscale,
                                      float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                                      const float *d_dout_dropped, int64_t lddo_dropped, float *d_dz0_amax, const uint32_t *d_row_idx,
                                      int dz_compact, int t_branch, float *d_t_out, int amax_branch, void *stream_) {
  return act_norm_bwd_impl(nb, d_Z, ldz, d_bias, act, d_scale, d_offset, n, F, seg, out_scale, d_dout, lddo, d_dZ, lddz, d_dscale, d_doffset,
                           d_dbias, d_partial, drop_p, drop_seed, d_dout_dropped, lddo_dropped, d_dz0_amax, d_row_idx, dz_compact, t_branch,
                           d_t_out, nullptr, amax_branch, stream_);
}

// ... with the output gradient given as a table: row i's gradient is d_dout[d_dout_map[i], :] (sl_pool_grad_table)
extern "C" int sl_act_norm_bwd_map(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                                   const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                                   uint32_t seg, float out_scale, const float *d_dout, int64_t lddo, const uint32_t *d_dout_map,
                                   float *const *d_dZ, const int64_t *lddz, float *d_dscale,
                                   float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                                   const float *d_dout_dropped, int64_t lddo_dropped, float *d_dz0_amax, void *stream_) {
  return act_norm_bwd_impl(nb, d_Z, ldz, d_bias, act, d_scale, d_offset, n, F, seg, out_scale, d_dout, lddo, d_dZ, lddz, d_dscale, d_doffset,
                           d_dbias, d_partial, drop_p, drop_seed, d_dout_dropped, lddo_dropped, d_dz0_amax, nullptr, 0, -1, nullptr,
                           d_dout_map, 0, stream_);
}
