// Shared host/device helpers for libshadow_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "shadow_hip.h"

namespace shadow {

// thread-local last-error string behind sg_last_error()
std::string &last_error();
int set_error(int code, const char *fmt, ...);

#define SHD_HIP(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return shadow::set_error(SG_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                 \
                               hipGetErrorString(_e), __FILE__, __LINE__);                 \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a driver round trip: remember the largest size granted per
// (kernel, device) and only ask again for more (the hot launches of a training step call it thousands of times otherwise)
hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes);

// 32-bit fill of a device buffer on a stream (prof.hip).  hipMemsetAsync's blit kernel takes 16 us for the 1.2 MB row-maximum /
// row-map arrays of a 289 k-row batch (rocprofv3: __amd_rocclr_fillBufferAligned, seven per headline step): a plain grid-stride
// store kernel fills them at HBM speed.  `value`: the word written (0xFFFFFFFF for the index arrays cleared to "none").
int fill_words(void *dst, uint32_t value, size_t nwords, hipStream_t st);

// Optional per-kernel timing inside the C entries (prof.hip): SHD_PROF(name, algorithmic bytes, flops, stream) at the top of
// a launch scope records a HIP-event pair around it while sl_prof_enable(1) is in effect; one branch otherwise.
bool prof_enabled();
struct ProfScope {
  int idx;
  uint32_t gen;          // generation of the record table the scope's slot belongs to (sl_prof_enable(1) starts a new one)
  hipStream_t st;
  ProfScope(const char *name, double bytes, double flops, hipStream_t st);
  ~ProfScope();
};
#define SHD_PROF(NAME, BYTES, FLOPS, ST) shadow::ProfScope prof_scope_((NAME), (double)(BYTES), (double)(FLOPS), (hipStream_t)(ST))
// (name built only when profiling is on)
#define SHD_PROF_FMT(BYTES, FLOPS, ST, ...)                                              \
  char prof_name_[96];                                                                   \
  prof_name_[0] = 0;                                                                     \
  if (shadow::prof_enabled()) snprintf(prof_name_, sizeof(prof_name_), __VA_ARGS__);    \
  SHD_PROF(prof_name_, BYTES, FLOPS, ST)

// fp16 weight images of the GEMM-epilogue kernels (gemm.hip; used by gemm_fused.hip)
struct PackF16Src {
  const float *B1, *B2;          // element (j, k) = B1[j s1j + k s1k] for k < K1, B2[j s2j + (k - K1) s2k] behind
  int64_t s1j, s1k, s2j, s2k;
  uint32_t K1;
  void *img;
  float *trailer;
};
size_t pack_f16_image_bytes(uint32_t K, uint32_t tiles);
size_t pack_f16_trailer_bytes(uint32_t tiles);
int pack_f16(int nimg, const PackF16Src *src, uint32_t N, uint32_t K, uint32_t tiles, float *zero, uint32_t n_zero, hipStream_t st);
// block-diagonal SpMM on line-pitched rows that also zero-fills the pad of every row's last line (aggregate.hip)
bool spmm_blockdiag_lines_ok(uint32_t F, const float *X, int64_t ldx, const float *Y, int64_t ldy);
int spmm_blockdiag_padded(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w, const uint32_t *d_edge_perm,
                          const float *d_row_scale, const float *d_col_scale, const float *d_X, int64_t ldx, float *d_Y, int64_t ldy,
                          uint32_t n, uint32_t F, const uint32_t *d_subg_node_off, const uint32_t *d_subg_edge_off, uint32_t num_subg,
                          uint32_t max_subg_nodes, float *d_row_amax, void *stream_);
// pipelined CSR SpMM with the output's row maxima from the same pass (aggregate.hip): whole rows per wavefront for 128 < F <= 256;
// amax_join: d_row_amax holds the maxima of other columns of the same operand
bool spmm_csr_whole_rows(uint32_t F, const float *X, int64_t ldx, const float *Y, int64_t ldy);
int spmm_csr_amax(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w, const uint32_t *d_edge_perm,
                  const float *d_row_scale, const float *d_col_scale, const float *d_X, int64_t ldx, float *d_Y, int64_t ldy, uint32_t n,
                  uint32_t F, float *d_row_amax, int amax_join, void *stream_);

constexpr int kWave = 64;  // CDNA wavefront

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }

// lanes below the calling lane
__device__ __forceinline__ uint64_t lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

// Inclusive scans across the 64 lanes of a wave on the DPP network (no LDS crossbar round trips): row_shr 1/2/4/8
// inside the 16-lane rows, then row_bcast:15 (lane 15 of a row into the next row) and row_bcast:31 (lane 31 into
// the upper half).  Lanes without a source keep the identity passed as `old`.
#define SHD_DPP_STEP(OP, CTRL, ROWMASK) v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ uint32_t dpp_add_(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t dpp_max_(uint32_t a, uint32_t b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  SHD_DPP_STEP(dpp_add_, 0x111, 0xf); SHD_DPP_STEP(dpp_add_, 0x112, 0xf); SHD_DPP_STEP(dpp_add_, 0x114, 0xf);
  SHD_DPP_STEP(dpp_add_, 0x118, 0xf); SHD_DPP_STEP(dpp_add_, 0x142, 0xa); SHD_DPP_STEP(dpp_add_, 0x143, 0xc);
  return v;
}

// inclusive max-scan of unsigned values (identity 0)
__device__ __forceinline__ uint32_t wave_incl_max_scan(uint32_t v) {
  SHD_DPP_STEP(dpp_max_, 0x111, 0xf); SHD_DPP_STEP(dpp_max_, 0x112, 0xf); SHD_DPP_STEP(dpp_max_, 0x114, 0xf);
  SHD_DPP_STEP(dpp_max_, 0x118, 0xf); SHD_DPP_STEP(dpp_max_, 0x142, 0xa); SHD_DPP_STEP(dpp_max_, 0x143, 0xc);
  return v;
}
#undef SHD_DPP_STEP

// ---------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_reduce_sum_f(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ uint32_t wave_reduce_max(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint32_t t = __shfl_xor(v, off, 64);
    v = t > v ? t : v;
  }
  return v;
}

// Block-wide exclusive scan of one value per thread.  `wsum` is LDS scratch of
// at least blockDim.x/64 + 1 words.  Returns the exclusive prefix; *total gets
// the block sum.  Contains two __syncthreads().
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum, uint32_t *total) {
  const uint32_t lane = lane_id(), wave = wave_id();
  const uint32_t nw = (blockDim.x + 63) >> 6;
  uint32_t incl = wave_incl_scan(v);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t w = 0; w < nw; w++) {
    uint32_t x = wsum[w];
    if (w < wave) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// Two block-wide exclusive scans sharing their barriers.  `wsum`: 2 * (blockDim.x / 64) + 2 words of LDS scratch.
__device__ __forceinline__ uint32_t block_excl_scan2(uint32_t va, uint32_t vb, uint32_t *wsum, uint32_t *total_a, uint32_t *excl_b,
                                                     uint32_t *total_b) {
  const uint32_t lane = lane_id(), wave = wave_id();
  const uint32_t nw = (blockDim.x + 63) >> 6;
  const uint32_t ia = wave_incl_scan(va), ib = wave_incl_scan(vb);
  if (lane == 63) { wsum[wave] = ia; wsum[nw + wave] = ib; }
  __syncthreads();
  uint32_t base_a = 0, tot_a = 0, base_b = 0, tot_b = 0;
  for (uint32_t w = 0; w < nw; w++) {
    const uint32_t xa = wsum[w], xb = wsum[nw + w];
    if (w < wave) { base_a += xa; base_b += xb; }
    tot_a += xa; tot_b += xb;
  }
  __syncthreads();
  *total_a = tot_a; *total_b = tot_b;
  *excl_b = base_b + ib - vb;
  return base_a + ia - va;
}

}  // namespace shadow
