// top_bwd.hip -- row sets and the input gradient of the EXACT row-sparse backward pass of a stack's top layer.
//
// Residue 'none' + centre pooling on a node task reads ONE row per subgraph of the last layer's output -- the root's
// (shaDow/layers.py:159-163: feats_in_l[-1][idx_targets]) -- so the gradient of that output is zero outside the roots R,
// the top GraphSAGE layer's dZs / dZn are zero outside R, and its input gradient
//     dX = dZs Ws + A^T (dZn Wn)
// is zero outside T = R u N(R).  Every subgraph has ONE root, so a row j of T receives at most one neighbour term -- from
// its own subgraph's root r, through the edge (r, j) -- plus the self term when j is the root itself:
//     dX[j] = A[r, j] . (dZn[r] Wn)  +  [j == r] . (dZs[r] Ws).
// sl_top_plan lists T (ascending, subgraph by subgraph: the batch is block diagonal and the roots ascend with the
// subgraphs) with, per row, its root's slot and the position of the edge (r, j) in the batch CSR; sl_top_dx forms dX on
// those rows.  The forward pass is not touched: every row of every layer is computed, as in the reference.
#include <algorithm>

#include "common.h"

namespace shadow {
namespace {

// One wavefront per subgraph: t_s = deg(r_s) + [r_s not among its own neighbours] -> off[s] (scanned below); a neighbour list
// that is not strictly ascending (repeated edges), or row sets of consecutive subgraphs that interleave, raise off[P + 1]
__global__ void __launch_bounds__(256) top_count_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                        const uint32_t *__restrict__ targets, uint32_t P, uint32_t *__restrict__ off) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= P) return;
  const uint32_t r = targets[s], e0 = indptr[r], e1 = indptr[r + 1];
  uint32_t self = 0, bad = 0;
  for (uint32_t e = e0 + lane; e < e1; e += 64) {
    const uint32_t c = indices[e];
    self |= c == r ? 1u : 0u;
    bad |= (e > e0 && c <= indices[e - 1]) ? 1u : 0u;
  }
  for (int d = 32; d >= 1; d >>= 1) { self |= __shfl_xor(self, d, 64); bad |= __shfl_xor(bad, d, 64); }
  if (lane == 0) {
    // T must come out strictly ascending over the whole batch (every row of T then receives exactly one neighbour term and
    // rowmap[T] = arange has no collisions): this subgraph's smallest row lies above the previous subgraph's largest.  True for
    // any collated batch (block-diagonal CSR, one root per block); a hand-made batch whose targets share a neighbour, or whose
    // target sits outside its own block, raises the flag instead of producing overwritten terms
    if (s > 0) {
      const uint32_t rp = targets[s - 1], p0 = indptr[rp], p1 = indptr[rp + 1];
      const uint32_t pmax = p1 > p0 ? max(rp, indices[p1 - 1]) : rp;
      const uint32_t mymin = e1 > e0 ? min(r, indices[e0]) : r;
      if (mymin <= pmax) bad = 1u;
    }
    off[s] = (e1 - e0) + (self ? 0u : 1u);
    if (bad) off[P + 1] = 1u;                               // (cleared by the host entry; the caller falls back to the dense pass)
  }
}

// One workgroup: exclusive scan of off[0 .. P) in place, off[P] = total
__global__ void __launch_bounds__(1024) top_scan_kernel(uint32_t P, uint32_t *__restrict__ off) {
  __shared__ uint32_t cell[1024];
  __shared__ uint32_t carry;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < P; base += 1024) {
    const uint32_t s = base + tid;
    const uint32_t cnt = s < P ? off[s] : 0u;
    cell[tid] = cnt;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      const uint32_t v = tid >= d ? cell[tid - d] : 0u;
      __syncthreads();
      cell[tid] += v;
      __syncthreads();
    }
    if (s < P) off[s] = carry + cell[tid] - cnt;
    __syncthreads();
    if (tid == 1023) carry += cell[1023];
    __syncthreads();
  }
  if (tid == 0) off[P] = carry;
}

// One wavefront per subgraph: merge the root into its (ascending) neighbour list
__global__ void __launch_bounds__(256) top_fill_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                       const uint32_t *__restrict__ targets, uint32_t P, const uint32_t *__restrict__ off,
                                                       uint32_t cap, uint32_t *__restrict__ T, uint32_t *__restrict__ slot,
                                                       int32_t *__restrict__ epos, uint32_t *__restrict__ self_idx) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= P) return;
  const uint32_t r = targets[s], e0 = indptr[r], e1 = indptr[r + 1], o = off[s];
  if (off[P] > cap) return;                              // (the caller sees the count and grows)
  // neighbours below the root keep their place, the root comes next (once), the others move up by one unless the root was
  // among them
  uint32_t below = 0, has = 0;
  for (uint32_t e = e0 + lane; e < e1; e += 64) { const uint32_t c = indices[e]; below += c < r ? 1u : 0u; has |= c == r ? 1u : 0u; }
  for (int d = 32; d >= 1; d >>= 1) { below += __shfl_xor(below, d, 64); has |= __shfl_xor(has, d, 64); }
  for (uint32_t e = e0 + lane; e < e1; e += 64) {
    const uint32_t c = indices[e];
    const uint32_t k = (e - e0) + ((c > r && !has) ? 1u : 0u);
    T[o + k] = c; slot[o + k] = s; epos[o + k] = (int32_t)e;
  }
  if (lane == 0) {
    self_idx[s] = o + below;
    if (!has) { T[o + below] = r; slot[o + below] = s; epos[o + below] = -1; }
  }
}

// ---- the batch's transposed adjacency restricted to the columns T (round 5) -----------------------------------------------
// A^T dZn of the layer below the row-sparse pass has terms from the rows T only: its aggregation then walks ~300 of a
// subgraph's ~2 000 transposed edges.  rowmap[i] = position of row i in T (all ones: not in T); per row of A^T the kept
// entries in their original order -- the same sums, bit for bit, as the walk over zero rows -- as a CSR whose column ids are
// positions in the compact [t, F] gradient, f_perm = the edge's place in the batch CSR (its drop-edge weight).
__global__ void __launch_bounds__(256) top_rowmap_kernel(const uint32_t *__restrict__ T, const uint32_t *__restrict__ off, uint32_t P, uint32_t cap,
                                                         uint32_t n, uint32_t *__restrict__ rowmap) {
  const uint32_t t = min(off[P], cap);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < t; i += gridDim.x * 256) { const uint32_t r = T[i]; if (r < n) rowmap[r] = i; }
}

// thread per row j of A^T: entries whose source row lies in T
__global__ void __launch_bounds__(256) top_filt_count_kernel(const uint32_t *__restrict__ t_indptr, const uint32_t *__restrict__ t_indices,
                                                             const uint32_t *__restrict__ rowmap, uint32_t n, uint32_t *__restrict__ cnt) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t c = 0;
  for (uint32_t p = t_indptr[j]; p < t_indptr[j + 1]; p++) c += rowmap[t_indices[p]] != 0xFFFFFFFFu ? 1u : 0u;
  cnt[j] = c;
}

// exclusive scan of cnt[0 .. n) in three steps: 1 024-element blocks (local prefix in place, block total to bsum), the block totals
// by top_scan_kernel, then the block's offset added; f_indptr[n] = total
__global__ void __launch_bounds__(1024) top_scan_local_kernel(uint32_t *__restrict__ v, uint32_t n, uint32_t *__restrict__ bsum) {
  __shared__ uint32_t wsum[32];
  const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
  const uint32_t x = i < n ? v[i] : 0u;
  uint32_t tot;
  const uint32_t ex = block_excl_scan(x, wsum, &tot);
  if (i < n) v[i] = ex;
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) top_scan_add_kernel(uint32_t *__restrict__ v, uint32_t n, const uint32_t *__restrict__ bsum, uint32_t nblocks,
                                                            uint32_t *__restrict__ total_out) {
  const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
  if (i < n) v[i] += bsum[blockIdx.x];
  if (i == 0) { v[n] = bsum[nblocks]; *total_out = bsum[nblocks]; }
}

__global__ void __launch_bounds__(256) top_filt_fill_kernel(const uint32_t *__restrict__ t_indptr, const uint32_t *__restrict__ t_indices,
                                                            const uint32_t *__restrict__ t_perm, const uint32_t *__restrict__ rowmap, uint32_t n,
                                                            const uint32_t *__restrict__ f_indptr, uint32_t *__restrict__ f_indices,
                                                            uint32_t *__restrict__ f_perm) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t o = f_indptr[j];
  for (uint32_t p = t_indptr[j]; p < t_indptr[j + 1]; p++) {
    const uint32_t m = rowmap[t_indices[p]];
    if (m != 0xFFFFFFFFu) { f_indices[o] = m; f_perm[o] = t_perm[p]; o++; }
  }
}

// dX[j, :] = w(j) G[slot(j), :] + [j is its subgraph's root] S[slot(j), :];  F % 4 == 0, a row on F / 4 lanes
__global__ void __launch_bounds__(256) top_dx_kernel(const float *__restrict__ G, const float *__restrict__ S, int64_t ldg, const uint32_t *__restrict__ T,
                                                     const uint32_t *__restrict__ slot, const int32_t *__restrict__ epos,
                                                     const uint32_t *__restrict__ self_idx, const uint32_t *__restrict__ targets,
                                                     const float *__restrict__ edge_w, const float *__restrict__ row_scale,
                                                     const float *__restrict__ col_scale, uint32_t t, uint32_t F, float *__restrict__ out, int64_t ldo) {
  const uint32_t f4 = F >> 2;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < (uint64_t)t * f4; i += (uint64_t)gridDim.x * 256) {
    const uint32_t j = (uint32_t)(i / f4), c = (uint32_t)(i % f4) * 4;
    const uint32_t s = slot[j];
    const int32_t e = epos[j];
    float w = 0.f;
    if (e >= 0) {
      w = edge_w ? edge_w[e] : 1.0f;
      if (row_scale) w *= row_scale[targets[s]];
      if (col_scale) w *= col_scale[T[j]];
    }
    const float4 g = *reinterpret_cast<const float4 *>(G + (int64_t)s * ldg + c);
    float4 v = make_float4(w * g.x, w * g.y, w * g.z, w * g.w);
    if (self_idx[s] == j) {
      const float4 q = *reinterpret_cast<const float4 *>(S + (int64_t)s * ldg + c);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *reinterpret_cast<float4 *>(out + (int64_t)j * ldo + c) = v;
  }
}

}  // namespace
}  // namespace shadow

using namespace shadow;

extern "C" int sl_top_plan(const uint32_t *d_indptr, const uint32_t *d_indices, const uint32_t *d_targets, uint32_t num_subg, uint32_t cap,
                           uint32_t *d_off, uint32_t *d_T, uint32_t *d_slot, int32_t *d_epos, uint32_t *d_self_idx, void *stream) {
  if (num_subg == 0) return SG_OK;
  // (d_indices may be NULL: a batch without a single edge -- every root row is empty and nothing is read through it)
  if (!d_indptr || !d_targets || !d_off || !d_T || !d_slot || !d_epos || !d_self_idx)
    return set_error(SG_ERR_INVALID, "sl_top_plan: null argument");
  hipStream_t st = (hipStream_t)stream;
  SHD_HIP(hipMemsetAsync(d_off + num_subg + 1, 0, 4, st));
  hipLaunchKernelGGL(top_count_kernel, dim3((num_subg + 3) / 4), dim3(256), 0, st, d_indptr, d_indices, d_targets, num_subg, d_off);
  hipLaunchKernelGGL(top_scan_kernel, dim3(1), dim3(1024), 0, st, num_subg, d_off);
  hipLaunchKernelGGL(top_fill_kernel, dim3((num_subg + 3) / 4), dim3(256), 0, st, d_indptr, d_indices, d_targets, num_subg, d_off, cap, d_T, d_slot,
                     d_epos, d_self_idx);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

// Row map + filtered transposed structure behind sl_top_plan (same stream, no host round trip in between): d_rowmap [n],
// d_f_indptr [n + 1], d_f_indices / d_f_perm [e] (capacity: every edge), d_work [n / 1024 + 4] words; the number of kept
// entries lands in d_off[num_subg + 2] beside the plan's own counters.  The kernels run whatever the plan's outcome -- on
// min(d_off[num_subg], cap) entries of T: a plan that overflowed `cap` or raised its flag leaves unusable structures behind, and
// the caller, who reads both counters with the same copy, drops them with the plan.
extern "C" int sl_top_plan_filter(const uint32_t *d_T, uint32_t *d_off, uint32_t num_subg, uint32_t cap, const uint32_t *d_t_indptr,
                                  const uint32_t *d_t_indices, const uint32_t *d_t_perm, uint32_t n, uint32_t *d_rowmap,
                                  uint32_t *d_f_indptr, uint32_t *d_f_indices, uint32_t *d_f_perm, uint32_t *d_work, void *stream) {
  if (n == 0 || num_subg == 0) return SG_OK;
  if (!d_T || !d_off || !d_t_indptr || !d_rowmap || !d_f_indptr || !d_f_indices || !d_f_perm || !d_work)
    return set_error(SG_ERR_INVALID, "sl_top_plan_filter: null argument");
  hipStream_t st = (hipStream_t)stream;
  { const int frc = fill_words(d_rowmap, 0xFFFFFFFFu, (size_t)n, st); if (frc != SG_OK) return frc; }
  hipLaunchKernelGGL(top_rowmap_kernel, dim3(std::min<uint32_t>((cap + 255) / 256, 2048)), dim3(256), 0, st, d_T, d_off, num_subg, cap, n, d_rowmap);
  hipLaunchKernelGGL(top_filt_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_t_indptr, d_t_indices, d_rowmap, n, d_f_indptr);
  const uint32_t nblocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(top_scan_local_kernel, dim3(nblocks), dim3(1024), 0, st, d_f_indptr, n, d_work);
  hipLaunchKernelGGL(top_scan_kernel, dim3(1), dim3(1024), 0, st, nblocks, d_work);
  hipLaunchKernelGGL(top_scan_add_kernel, dim3(nblocks), dim3(1024), 0, st, d_f_indptr, n, d_work, nblocks, d_off + num_subg + 2);
  hipLaunchKernelGGL(top_filt_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_t_indptr, d_t_indices, d_t_perm, d_rowmap, n, d_f_indptr,
                     d_f_indices, d_f_perm);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_top_dx(const float *d_G, const float *d_S, int64_t ldg, const uint32_t *d_T, const uint32_t *d_slot, const int32_t *d_epos,
                         const uint32_t *d_self_idx, const uint32_t *d_targets, const float *d_edge_w, const float *d_row_scale,
                         const float *d_col_scale, uint32_t t, uint32_t F, float *d_out, int64_t ldo, void *stream) {
  if (t == 0 || F == 0) return SG_OK;
  if (!d_G || !d_S || !d_T || !d_slot || !d_epos || !d_self_idx || !d_targets || !d_out) return set_error(SG_ERR_INVALID, "sl_top_dx: null argument");
  if ((F & 3) || (ldg & 3) || (ldo & 3) || (reinterpret_cast<uintptr_t>(d_G) & 15) || (reinterpret_cast<uintptr_t>(d_S) & 15) ||
      (reinterpret_cast<uintptr_t>(d_out) & 15))
    return set_error(SG_ERR_INVALID, "sl_top_dx: needs F %% 4 == 0 and 16-byte aligned rows");
  const uint64_t items = (uint64_t)t * (F >> 2);
  const uint32_t grid = (uint32_t)std::min<uint64_t>((items + 255) / 256, 256 * 8);
  SHD_PROF_FMT(4.0 * t * F + 12.0 * t, 0, stream, "top_dx_F%u", F);
  hipLaunchKernelGGL(top_dx_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_G, d_S, ldg, d_T, d_slot, d_epos, d_self_idx, d_targets,
                     d_edge_w, d_row_scale, d_col_scale, t, F, d_out, ldo);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}
