// gemm_fused.hip -- the split-bf16 MFMA GEMM of gemm.hip with the layer's activation + feature normalisation riding in
// its epilogue (shaDow/layers.py:329-338 `_f_norm_feat`, :434-435 GCN, :476-483 GraphSAGE).
//
// A wavefront of gemm_nt_split_kernel owns 32 rows x ALL N <= 256 output columns, i.e. whole feature rows -- exactly the
// unit the row normalisation works on.  Two fusions are built on that:
//
//   forward  (MODE 0)   Z_b = A_b . W_b^T for the nb <= 2 branches of a layer in ONE launch (the k-step sequence of the
//                       second product simply continues the first; the B-image ring never drains), Z_b written once, and
//                           out = out_scale * sum_b norm_b(act(Z_b + bias_b))   [+ the next layer's input dropout]
//                       produced from the accumulators: removes the act_norm forward launch and its re-read of every Z_b
//                       (2 x 4 n F bytes per GraphSAGE layer).
//   backward (MODE 1)   dOut = [dZs | A^T dZn] . [Ws ; Wn] of layer l is the gradient of layer l-1's output; instead of
//                       writing it and launching act_norm backward on it, the epilogue reads layer l-1's saved Z_b and
//                       writes dZ_b of layer l-1 directly (+ the per-workgroup partial sums of dscale / doffset / dbias):
//                       removes one [n, F] write, one [n, F] read and a launch per layer boundary.
//
// Epilogue mechanics: after the last k-step the B ring in LDS is dead.  Each wavefront parks its 32 x N accumulator tile
// there 16 rows at a time (C/D layout -> row-major, conflict-free ds_write_b32) and then walks the rows TWO PER PASS, one
// row on 32 lanes (a lane holds the float4s j, j + 32 of its row): the arithmetic of act_norm_kernel (aggregate.hip)
// with the row statistics as DPP butterflies + one v_permlane16_swap -- no dependent LDS-crossbar chain, independent
// float4s per lane -- and every global access of the epilogue a run of 512 contiguous bytes per row.  (First
// built with a whole wavefront per row and 64-lane shuffle sums: 4-8 dependent 6-step ds_bpermute chains per row made the
// epilogue longer than the main loop, products step 10.06 -> 11.30 ms.)
// The other workgroup resident on the CU keeps the matrix cores busy meanwhile (two 4-wave workgroups per CU).
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "actnorm_common.h"
#include "common.h"
#include "gat_act.h"
#include "gemm_common.h"

namespace shadow {
namespace {

struct FusedDesc {
  // GEMM: C[M, N] = A_p[M, K] . B_p[N, K]^T for the phases p < NBP; the B images lie back to back
  const float *A[2];
  int64_t lda[2];
  uint32_t asplit;                  // plain single product (MODE 2, one phase): units [asplit, units) of A come from A[1] (K-concatenated operand in two tensors); = units otherwise
  const float *cbias[2];            // MODE 2: per-column bias added to product b when it leaves (may be NULL)
  const float *aamax[2];            // largest magnitude of every row of A_p, [M]: the power-of-two row scale follows from it (gemm_common.h)
  const half8 *Bimg;                // the NBP fp16 images back to back ...
  const float *btrail[2];           // ... and the trailer of each: [32 TW] column scales, [32 TW] inverses
  uint32_t M, N, K, units;          // units = ceil(K / 32) per phase
  // the act + norm the epilogue applies (forward: to this GEMM's own outputs; backward: of the layer below)
  const float *bias[2];
  int act[2];
  const float *scale, *offset;      // [nb, N]
  float out_scale, eps;
  uint32_t drop_thr;                // fused output dropout (rule of sl_act_norm_fwd); 0: none
  float drop_scale;
  uint32_t seed_lo, seed_hi;
  // forward
  float *Z[2];                      // pre-activations (without bias), written by the kernel
  int64_t ldz[2];
  float *out;  int64_t ldo;
  float *out2; int64_t ldo2;        // dual mode: out stays plain, out2 receives the dropped values
  float *out_amax;                  // optional: row maxima of out2 (or out) for the GEMM that reads it next
  float *stats_w;                   // optional [M, 2 NBA]: (mean, 1 / std) of act(Z_b + bias_b) per row and branch, for the backward epilogue
  // backward: the layer below
  const float *Zr[2]; int64_t ldzr[2];
  float *dZ[2];       int64_t lddz[2];
  float *partial;                   // [grid, nb, 3, N]
  float *dz_amax;                   // optional: row maxima of dZ[0] (the left half of the next K = 2F operand)
  const float *stats_r;             // optional [M, 2 NBA]: the row statistics the forward epilogue left (stats_w); NULL: recomputed
  // backward, optional: a row-sparse ADDEND of the product -- row i of G gets corr[corr_row[i], :] added before the epilogue
  // uses it when corr_row[i] < corr_rows (G = A B^T + scatter(corr): the part of a K-concatenated product whose operand rows
  // are non-zero on a few rows only, computed on those rows by the caller)
  const float *corr; int64_t ldcorr;
  const uint32_t *corr_row; uint32_t corr_rows;
  // backward, MODE -1: a DENSE addend of the lower layer's output gradient that does NOT pass the dropout mask -- the layer
  // below is in dual-output mode (its plain output feeds a read-out, its dropped output the layer above): the epilogue forms
  //   dy = G * mask / (1 - p)  +  dplain * out_scale        (G: this product = the gradient of the dropped output)
  // before the act + norm backward; what sl_act_norm_bwd does with (d_dout, d_dout_dropped)
  const float *dplain; int64_t lddplain;
  // ... or, with plain_row set, row i's addend is dplain[plain_row[i], :]: the gradient of a mean / sum pooling read-out is one row
  // per subgraph (+ one per root), kept as that small table instead of its [M, N] expansion (sl_pool_grad_table)
  const uint32_t *plain_row;
  // MODE 2, optional: the GAT layer's per-node terms from the paired Linear's own tiles (sl_gemm_nt2_gat_f32).  For product b with
  // gat_u[b] set: u[row, h] = sum over head h's D columns of gat_att[b][col] * act(C[row, col]) -- shaDow/layers.py:566-569 -- and,
  // with gat_store_act[b], the tile leaves as act(C) (hn = act(z_neigh): the pre-activation is never written).  N == 32 TW == H D.
  const float *gat_att[2];
  float *gat_u[2];
  int gat_store_act[2];
  int gat_act;
  uint32_t gat_ls, gat_H;           // lanes per head slice (D / 4: a power of two <= 32), heads
};

// The epilogue puts ONE FEATURE ROW ON 32 LANES (2 rows per wavefront pass; 64 lanes in the backward form): the row
// statistics are group_sum<LPR> (actnorm_common.h: DPP butterflies + permlane swaps, five or six VALU operations deep),
// every lane carries Q independent float4s, and the column accumulators of the backward form (5 Q float4s per lane) stay
// inside the register budget of two wavefronts per SIMD.
template <int LPR> __device__ __forceinline__ float row_sum(float v) { return group_sum<LPR>(v); }

__device__ __forceinline__ float hsum4(const float4 &v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4sel(bool c, const float4 &v) { return c ? v : f4zero(); }

// the per-column parameters of one lane (constant over the rows: loaded once per epilogue)
template <int NBA, int Q>
struct ColParams {
  float4 bias[NBA][Q], sc[NBA][Q], of[NBA][Q];
};

// keep-factors of the fused dropout for the float4 at column c of a row whose hash base is rowh (mask rule: actnorm_common.h)
__device__ __forceinline__ float4 drop_factors(uint32_t rowh, uint32_t c, uint32_t thr, float keep_value) {
  const uint32_t keep = drop_keep4_h(rowh, thr, c);
  return make_float4((keep & 1u) ? keep_value : 0.f, (keep & 2u) ? keep_value : 0.f, (keep & 4u) ? keep_value : 0.f, (keep & 8u) ? keep_value : 0.f);
}

// One row on 32 lanes: lane j holds the float4s j, j + 32, ... of the row (columns 4 (j + 32 q) ..).
// out[q] = sum_b norm_b(act(z_b[q] + bias_b))   (before out_scale); on[q]: the float4 lies inside the row.
// ACT >= 0: the activation of every branch, known at compile time; ACT < 0: d.act[b] at run time.
template <int NBA, int Q, int ACT, int LPR>
__device__ __forceinline__ void an_rows_fwd(const FusedDesc &d, const ColParams<NBA, Q> &cp, const bool (&on)[Q], const float4 (&zc)[NBA][Q],
                                            float inv_seg, float4 (&out)[Q], float *stat_row) {
#pragma unroll
  for (int q = 0; q < Q; q++) out[q] = f4zero();
#pragma unroll
  for (int b = 0; b < NBA; b++) {
    const int act = ACT >= 0 ? ACT : d.act[b];
    float4 h[Q];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const float4 z = make_float4(zc[b][q].x + cp.bias[b][q].x, zc[b][q].y + cp.bias[b][q].y, zc[b][q].z + cp.bias[b][q].z,
                                   zc[b][q].w + cp.bias[b][q].w);
      h[q] = f4sel(on[q], make_float4(act_fwd(act, z.x), act_fwd(act, z.y), act_fwd(act, z.z), act_fwd(act, z.w)));
      s += hsum4(h[q]);
    }
    // biased mean / variance over the row (layers.py:334-335)
    const float mean = row_sum<LPR>(s) * inv_seg;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      h[q] = f4sel(on[q], make_float4(h[q].x - mean, h[q].y - mean, h[q].z - mean, h[q].w - mean));
      v += (h[q].x * h[q].x + h[q].y * h[q].y) + (h[q].z * h[q].z + h[q].w * h[q].w);
    }
    const float rstd = rsqrtf(row_sum<LPR>(v) * inv_seg + d.eps);
    if (stat_row) { stat_row[2 * b] = mean; stat_row[2 * b + 1] = rstd; }     // (one lane of the row; nullptr elsewhere)
#pragma unroll
    for (int q = 0; q < Q; q++) {
      // (x - mean) * scale * rsqrt(var) + offset   (layers.py:336)
      out[q].x += h[q].x * cp.sc[b][q].x * rstd + cp.of[b][q].x; out[q].y += h[q].y * cp.sc[b][q].y * rstd + cp.of[b][q].y;
      out[q].z += h[q].z * cp.sc[b][q].z * rstd + cp.of[b][q].z; out[q].w += h[q].w * cp.sc[b][q].w * rstd + cp.of[b][q].w;
    }
  }
}

// act_norm backward of one row on 32 lanes: dy[q] = gradient of the row's output, already through the dropout mask and
// zero outside the row / the matrix; writes dZ_b, accumulates the column sums of dscale (gs), doffset (go) and dbias (gb)
// kStats: (mean, 1 / std) of every branch come in `st` (what the forward epilogue stored) instead of two row reductions each
template <int NBA, int Q, int ACT, int LPR, bool kStats>
__device__ __forceinline__ void an_rows_bwd(const FusedDesc &d, uint64_t row, bool row_ok, uint32_t j, const bool (&on)[Q],
                                            const float4 (&dy)[Q], const float4 (&zc)[NBA][Q], float inv_seg, float4 (&gs)[NBA][Q],
                                            float4 (&go)[Q], float4 (&gb)[NBA][Q], const float (&st)[2 * NBA]) {
#pragma unroll
  for (int q = 0; q < Q; q++) { go[q].x += dy[q].x; go[q].y += dy[q].y; go[q].z += dy[q].z; go[q].w += dy[q].w; }
#pragma unroll
  for (int b = 0; b < NBA; b++) {
    const int act = ACT >= 0 ? ACT : d.act[b];
    // (the per-column parameters come from the L1 here: with the 5 Q column accumulators there is no room to keep them)
    float4 z[Q], h[Q], xh[Q], dxh[Q];
    float s = 0.f, zmax = 0.f;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const float4 bb = d.bias[b] ? ld4(d.bias[b] + (on[q] ? 4 * (j + LPR * q) : 0u)) : f4zero();
      z[q] = make_float4(zc[b][q].x + bb.x, zc[b][q].y + bb.y, zc[b][q].z + bb.z, zc[b][q].w + bb.w);
      h[q] = f4sel(on[q], make_float4(act_fwd(act, z[q].x), act_fwd(act, z[q].y), act_fwd(act, z[q].z), act_fwd(act, z[q].w)));
      s += hsum4(h[q]);
    }
    float mean, rstd;
    if (kStats) {
      mean = st[2 * b]; rstd = st[2 * b + 1];
#pragma unroll
      for (int q = 0; q < Q; q++) xh[q] = f4sel(on[q], make_float4(h[q].x - mean, h[q].y - mean, h[q].z - mean, h[q].w - mean));
      (void)s;
    } else {
      mean = row_sum<LPR>(s) * inv_seg;
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < Q; q++) {
        xh[q] = f4sel(on[q], make_float4(h[q].x - mean, h[q].y - mean, h[q].z - mean, h[q].w - mean));
        v += (xh[q].x * xh[q].x + xh[q].y * xh[q].y) + (xh[q].z * xh[q].z + xh[q].w * xh[q].w);
      }
      rstd = rsqrtf(row_sum<LPR>(v) * inv_seg + d.eps);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      xh[q].x *= rstd; xh[q].y *= rstd; xh[q].z *= rstd; xh[q].w *= rstd;
      gs[b][q].x += dy[q].x * xh[q].x; gs[b][q].y += dy[q].y * xh[q].y; gs[b][q].z += dy[q].z * xh[q].z; gs[b][q].w += dy[q].w * xh[q].w;
      const float4 sc = ld4(d.scale + (size_t)b * d.N + (on[q] ? 4 * (j + LPR * q) : 0u));
      dxh[q] = make_float4(dy[q].x * sc.x, dy[q].y * sc.y, dy[q].z * sc.z, dy[q].w * sc.w);
      s1 += hsum4(dxh[q]);
      s2 += (dxh[q].x * xh[q].x + dxh[q].y * xh[q].y) + (dxh[q].z * xh[q].z + dxh[q].w * xh[q].w);
    }
    const float m1 = row_sum<LPR>(s1) * inv_seg, m2 = row_sum<LPR>(s2) * inv_seg;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      float4 dh = make_float4(rstd * (dxh[q].x - m1 - xh[q].x * m2), rstd * (dxh[q].y - m1 - xh[q].y * m2),
                              rstd * (dxh[q].z - m1 - xh[q].z * m2), rstd * (dxh[q].w - m1 - xh[q].w * m2));
      dh.x *= act_bwd(act, z[q].x, h[q].x); dh.y *= act_bwd(act, z[q].y, h[q].y);
      dh.z *= act_bwd(act, z[q].z, h[q].z); dh.w *= act_bwd(act, z[q].w, h[q].w);
      if (row_ok && on[q]) st4s(d.dZ[b] + row * d.lddz[b] + 4 * (j + LPR * q), dh);
      gb[b][q].x += dh.x; gb[b][q].y += dh.y; gb[b][q].z += dh.z; gb[b][q].w += dh.w;
      if (b == 0) zmax = fmaxf(zmax, amax4(dh));
    }
    if (b == 0 && d.dz_amax) {
      zmax = group_max<LPR>(zmax);
      if (j == 0 && row_ok) d.dz_amax[row] = zmax;
    }
  }
}

// Everything behind the last k-step, for one (activation, full-width) specialisation: the accumulator tile goes through
// the dead B ring 16 rows at a time and leaves two rows per pass (see the file header).
template <int TW, int MODE, int NBA, int ACT, bool kFull, bool kStats = false>
__device__ __forceinline__ void fused_epilogue(const FusedDesc &d, f32x16 (&acc)[TW], unsigned char *gsm, uint64_t m0) {
  constexpr int SP = 32 * TW;                               // row pitch of the stash (floats)
  // forward: a row on 32 lanes (two rows per pass, Q = TW / 4 float4s per lane); backward: one float4 per lane (a 256-wide
  // row on the whole wavefront) -- its 5 Q column accumulators have to fit beside the waiting half of the tile
  constexpr bool kBwd = MODE == 1 || MODE == -1, kPlain = MODE == -1;
  constexpr int LPR = (kBwd && TW == 8) ? 64 : 32;
  constexpr int RP = 64 / LPR;                              // rows per pass
  constexpr int Q = 8 * TW / LPR;                           // float4s per lane
  constexpr int kThreads = 256;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t r = lane & 31u, g = lane >> 5;             // C/D layout of the accumulators
  const uint32_t j = lane & (LPR - 1), rs = lane / LPR;     // epilogue layout: lane j of row slot rs
  const uint32_t M = d.M;
  float *stash = reinterpret_cast<float *>(gsm) + (size_t)wv * (16 * SP);
  bool on[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) on[q] = kFull || 4 * (j + LPR * q) < d.N;
  const float inv_seg = 1.0f / (float)d.N;
  ColParams<NBA, Q> cp;
#pragma unroll
  for (int b = 0; b < (MODE == 0 ? NBA : 0); b++)
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const uint32_t c = on[q] ? 4 * (j + LPR * q) : 0u;     // (lanes outside the row read column 0: harmless, never used)
      cp.bias[b][q] = d.bias[b] ? ld4(d.bias[b] + c) : f4zero();
      cp.sc[b][q] = ld4(d.scale + (size_t)b * d.N + c);
      cp.of[b][q] = ld4(d.offset + (size_t)b * d.N + c);
    }
  float4 gs[NBA][Q], go[Q], gb[NBA][Q];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    go[q] = f4zero();
#pragma unroll
    for (int b = 0; b < NBA; b++) gs[b][q] = gb[b][q] = go[q];
  }
  if (MODE == 0 && NBA == 2) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // Z_0 is re-read below by other lanes
  constexpr int NG = MODE == 0 ? NBA - 1 : NBA;             // operands that come from memory: Z_0 (forward) / both Z of the layer below
  // One half (16 rows) of the tile at a time.  While the first half is processed the second half still sits in 8 TW
  // accumulator registers; during the second half those are free, so its global operands are fetched DEEPER ahead (the
  // register allocation is the maximum over both halves either way).
  auto half = [&](auto HFc, auto Dc) {
    constexpr int hf = decltype(HFc)::value;
    constexpr int D = decltype(Dc)::value;
    // rows 16 hf .. 16 hf + 15 of the tile: i = 8 hf + ii -> local row (ii & 3) + 8 (ii >> 2) + 4 g
#pragma unroll
    for (int t = 0; t < TW; t++)
#pragma unroll
      for (int ii = 0; ii < 8; ii++) stash[((ii & 3) + 8 * (ii >> 2) + 4 * g) * SP + 32 * t + r] = acc[t][8 * hf + ii];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint64_t rbase = m0 + 16u * hf + rs;              // this lane's row in pass 0; + RP per pass
    // The global operands of a pass are loaded D passes ahead (a ring of D register sets, the pass loop unrolled by D):
    // with two wavefronts per SIMD the epilogue has little else to hide HBM latency behind -- one pass ahead left it
    // latency-bound (2 KB in flight per wavefront).  Rows past the end re-read the last row (never stored).
    constexpr int P = 16 / RP;                              // passes per half
#ifndef SHADOW_EPI_DEPTH_BWD
#define SHADOW_EPI_DEPTH_BWD 2
#endif
#ifndef SHADOW_EPI_DEPTH_FWD
#define SHADOW_EPI_DEPTH_FWD 1
#endif
    static_assert(P % D == 0, "");
    float4 zpre[D][NG > 0 ? NG : 1][Q];
    float spre[D][2 * NBA];                                 // (backward with kStats: the row's saved statistics, same look-ahead)
    uint32_t cpre[D];                                       // (backward with a sparse addend: its row of corr, or >= corr_rows)
    float4 ppre[D][kPlain ? Q : 1];                         // (backward with a dense plain addend: its row, same look-ahead)
    uint32_t pnext = (kPlain && d.plain_row) ? d.plain_row[min((uint64_t)rbase, (uint64_t)M - 1)] : 0u;
    uint32_t pnext2 = (kPlain && d.plain_row) ? d.plain_row[min((uint64_t)rbase + RP, (uint64_t)M - 1)] : 0u;
    auto load_pass = [&](int slot, uint64_t row) {
      const uint64_t rr = min(row, (uint64_t)M - 1);
      if (kBwd) cpre[slot] = d.corr ? d.corr_row[rr] : 0xFFFFFFFFu;
      if (kPlain) {
        if (d.plain_row) {                                  // (a few thousand distinct rows: cached loads)
          // (the map entries are asked for two passes earlier -- passes are loaded in row order, RP rows apart -- so that the table
          //  row's address is not a second round trip in front of this pass's loads)
          const float *pr = d.dplain + (uint64_t)pnext * d.lddplain;
          pnext = pnext2;
          pnext2 = d.plain_row[min(row + (uint64_t)(2 * RP), (uint64_t)M - 1)];
#pragma unroll
          for (int q = 0; q < Q; q++) ppre[slot][q] = ld4(pr + (on[q] ? 4 * (j + LPR * q) : 0u));
        } else {
#pragma unroll
          for (int q = 0; q < Q; q++) ppre[slot][q] = ld4s(d.dplain + rr * d.lddplain + (on[q] ? 4 * (j + LPR * q) : 0u));
        }
      }
      if (kBwd && kStats) {
        if (NBA == 2) {                                   // (one 16-byte load, the same address in every lane of the row)
          const float4 sv = ld4(d.stats_r + rr * 4);
          spre[slot][0] = sv.x; spre[slot][1] = sv.y; spre[slot][2 * NBA - 2] = sv.z; spre[slot][2 * NBA - 1] = sv.w;
        } else {
#pragma unroll
          for (int k = 0; k < 2 * NBA; k++) spre[slot][k] = d.stats_r[rr * (2 * NBA) + k];
        }
      }
#pragma unroll
      for (int b = 0; b < NG; b++)
#pragma unroll
        for (int q = 0; q < Q; q++) {
          const uint32_t c = on[q] ? 4 * (j + LPR * q) : 0u;
          if (MODE == 0) zpre[slot][b][q] = ld4(d.Z[0] + rr * d.ldz[0] + c);
          else zpre[slot][b][q] = ld4s(d.Zr[b] + rr * d.ldzr[b] + c);
        }
    };
#pragma unroll
    for (int k = 0; k < D; k++) load_pass(k, rbase + (uint32_t)(RP * k));
#pragma unroll 1
    for (int pg = 0; pg < P / D; pg++) {
#pragma unroll
    for (int k = 0; k < D; k++) {
      const int ps = pg * D + k;
      const uint64_t row = rbase + (uint32_t)(RP * ps);
      const bool row_ok = row < M;
      const float *srow = stash + (RP * ps + rs) * SP;
      float4 zc[NBA][Q];
      float4 own[Q];                                        // this GEMM's tile rows: Z_{NBA-1} (forward) / the output gradient (backward)
#pragma unroll
      for (int q = 0; q < Q; q++) own[q] = *reinterpret_cast<const float4 *>(srow + (on[q] ? 4 * (j + LPR * q) : 0u));
#pragma unroll
      for (int b = 0; b < NG; b++)
#pragma unroll
        for (int q = 0; q < Q; q++) zc[b][q] = zpre[k][b][q];
      float stc[2 * NBA];
#pragma unroll
      for (int kk = 0; kk < 2 * NBA; kk++) stc[kk] = (kBwd && kStats) ? spre[k][kk] : 0.f;
      float4 pl[kPlain ? Q : 1];
      if (kPlain) {
#pragma unroll
        for (int q = 0; q < Q; q++) pl[q] = ppre[k][q];
      }
      if (kBwd) {
        const uint32_t ci = cpre[k];
        if (ci < d.corr_rows) {                             // (one row in fourteen: the rows the sparse addend reaches)
#pragma unroll
          for (int q = 0; q < Q; q++) {
            if (on[q]) {
              const float4 cv = ld4(d.corr + (int64_t)ci * d.ldcorr + 4 * (j + LPR * q));
              own[q].x += cv.x; own[q].y += cv.y; own[q].z += cv.z; own[q].w += cv.w;
            }
          }
        }
      }
      if (pg + 1 < P / D) load_pass(k, row + (uint32_t)(RP * D));
      uint32_t rowh = 0;
      if (d.drop_thr) rowh = drop_row_hash(d.seed_lo, d.seed_hi, row);
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < Q; q++) zc[NBA - 1][q] = own[q];
        float4 o[Q];
        float omax = 0.f;
        an_rows_fwd<NBA, Q, ACT, LPR>(d, cp, on, zc, inv_seg, o, (d.stats_w && j == 0 && row_ok) ? d.stats_w + row * (2 * NBA) : nullptr);
#pragma unroll
        for (int q = 0; q < Q; q++) {
          if (row_ok && on[q]) {
            const uint32_t c = 4 * (j + LPR * q);
            st4s(d.Z[NBA - 1] + row * d.ldz[NBA - 1] + c, own[q]);
            float4 v = make_float4(o[q].x * d.out_scale, o[q].y * d.out_scale, o[q].z * d.out_scale, o[q].w * d.out_scale);
            if (d.drop_thr) {
              const float4 kf = drop_factors(rowh, c, d.drop_thr, d.drop_scale);
              const float4 dr = make_float4(v.x * kf.x, v.y * kf.y, v.z * kf.z, v.w * kf.w);
              if (d.out2) st4s(d.out2 + row * d.ldo2 + c, dr);      // dual mode: out stays un-dropped
              else v = dr;
              omax = fmaxf(omax, amax4(dr));
            } else {
              omax = fmaxf(omax, amax4(v));
            }
            st4s(d.out + row * d.ldo + c, v);
          }
        }
        if (d.out_amax) {                                  // the row's largest magnitude for the GEMM that reads it next
          omax = group_max<LPR>(omax);
          if (j == 0 && row_ok) d.out_amax[row] = omax;
        }
      } else {
        float4 dy[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
          // gradient of the fused output dropout: same mask, same 1 / (1 - p); nothing outside the matrix
          float4 dm = make_float4(d.out_scale, d.out_scale, d.out_scale, d.out_scale);
          if (d.drop_thr) dm = drop_factors(rowh, 4 * (j + LPR * q), d.drop_thr, d.out_scale * d.drop_scale);
          if (!(row_ok && on[q])) dm = f4zero();
          dy[q] = make_float4(own[q].x * dm.x, own[q].y * dm.y, own[q].z * dm.z, own[q].w * dm.w);
          if (kPlain && row_ok && on[q]) {          // (the plain output's gradient: no mask, sl_act_norm_bwd's dual-mode sum)
            dy[q].x += pl[q].x * d.out_scale; dy[q].y += pl[q].y * d.out_scale; dy[q].z += pl[q].z * d.out_scale; dy[q].w += pl[q].w * d.out_scale;
          }
        }
        if (!row_ok) {                                     // (a clamped row's Z must not reach the column sums either)
#pragma unroll
          for (int b = 0; b < NBA; b++)
#pragma unroll
            for (int q = 0; q < Q; q++) zc[b][q] = f4zero();
        }
        an_rows_bwd<NBA, Q, ACT, LPR, kStats>(d, row, row_ok, j, on, dy, zc, inv_seg, gs, go, gb, stc);
      }
      __builtin_amdgcn_sched_barrier(0);                    // (one pass at a time: interleaved passes multiply the live registers)
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // (measured, scripts/ab_fused_depth.sh, M = 289 k: deeper second halves spill -- backward 2/2 538 us, 2/4 587, 2/8 749;
  //  forward 1/1 420 us, 1/2 435, 1/4 437)
#ifndef SHADOW_EPI_DEPTH2_BWD
#define SHADOW_EPI_DEPTH2_BWD 2
#endif
#ifndef SHADOW_EPI_DEPTH2_FWD
#define SHADOW_EPI_DEPTH2_FWD 1
#endif
  constexpr int P_ = 16 / RP;
  constexpr int D0 = kBwd ? SHADOW_EPI_DEPTH_BWD : ((NBA == 2 && TW == 8) ? SHADOW_EPI_DEPTH_FWD : 2);
  constexpr int D1w = kBwd ? SHADOW_EPI_DEPTH2_BWD : ((NBA == 2 && TW == 8) ? SHADOW_EPI_DEPTH2_FWD : 2);
  constexpr int D1 = D1w > P_ ? P_ : D1w;
  half(std::integral_constant<int, 0>{}, std::integral_constant<int, D0>{});
  half(std::integral_constant<int, 1>{}, std::integral_constant<int, D1>{});
  if (kBwd) {
    // per-workgroup partial sums of the parameter gradients: RP row slots x 4 wavefronts column-sum rows in LDS, added
    // in a fixed order and left for act_norm_finish_kernel (deterministic)
    float *red = reinterpret_cast<float *>(gsm);            // [4 RP][1 + 2 NBA][SP]
    constexpr int kKinds = 1 + 2 * NBA;
    static_assert(4 * RP * kKinds * SP * 4 <= 4 * 16 * SP * 4, "the reduction rows must fit the stash allocation");
    __syncthreads();                                        // every wavefront is done with its stash
    float *rw = red + (size_t)(wv * RP + rs) * kKinds * SP;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      if (on[q]) {
        *reinterpret_cast<float4 *>(rw + 4 * (j + LPR * q)) = go[q];
#pragma unroll
        for (int b = 0; b < NBA; b++) {
          *reinterpret_cast<float4 *>(rw + (1 + 2 * b) * SP + 4 * (j + LPR * q)) = gs[b][q];
          *reinterpret_cast<float4 *>(rw + (2 + 2 * b) * SP + 4 * (j + LPR * q)) = gb[b][q];
        }
      }
    }
    __syncthreads();
    for (uint32_t c = tid; c < d.N; c += kThreads) {
      float sacc[kKinds];
#pragma unroll
      for (int k = 0; k < kKinds; k++) {
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4 * RP; w++) a4[w & 3] += red[((size_t)w * kKinds + k) * SP + c];
        sacc[k] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
      }
#pragma unroll
      for (int b = 0; b < NBA; b++) {
        float *pp = d.partial + (((size_t)blockIdx.x * NBA + b) * 3) * d.N + c;
        pp[0] = sacc[1 + 2 * b];                            // dscale
        pp[d.N] = sacc[0];                                  // doffset (the same sum for every branch)
        pp[2 * (size_t)d.N] = sacc[2 + 2 * b];              // dbias
      }
    }
  }
}

// The accumulators hold (scale_row A) . (scale_col B)^T: take both powers of two out again, in place.  Lane (r, g) of the
// C/D layout holds column 32 t + r of the rows (i & 3) + 8 (i >> 2) + 4 g; a row's inverse scale lives in lane `row`.
template <int TW>
__device__ __forceinline__ void unscale_tile(f32x16 (&acc)[TW], float ainv, const float *__restrict__ binv, uint32_t g, uint32_t r) {
  float cinv[TW];
#pragma unroll
  for (int t = 0; t < TW; t++) cinv[t] = binv[32 * t + r];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const uint32_t row = (i & 3) + 8 * (i >> 2) + 4 * g;
    const float rinv = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(4 * row), __float_as_int(ainv)));
#pragma unroll
    for (int t = 0; t < TW; t++) acc[t][i] *= rinv * cinv[t];
  }
}

// nn.Linear's bias on a plain product's tile (lane (r, g) holds column 32 t + r of tile t)
template <int TW>
__device__ __forceinline__ void add_bias_tile(f32x16 (&acc)[TW], const float *__restrict__ bias, uint32_t N, uint32_t r) {
#pragma unroll
  for (int t = 0; t < TW; t++) {
    const float bv = 32 * t + r < N ? bias[32 * t + r] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] += bv;
  }
}

// TW column tiles (N <= 32 TW); MODE 0 forward / 1 backward; NBP GEMM phases; NBA act_norm branches;
// kTail: K % 32 != 0 (the last unit of a phase is zero-padded).  Main loop = gemm_nt_split_kernel<1, TW, 1, 4>.
template <int TW, int MODE, int NBP, int NBA, bool kTail>
__global__ void __launch_bounds__(256, 2) gemm_nt_fused_kernel(const FusedDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  constexpr int kThreads = 256;
  constexpr int kStepVecs = 2 * TW * 64;                 // half8 vectors of one k-step's B image (pieces h, m)
  constexpr int kFill = (kStepVecs + kThreads - 1) / kThreads;
  constexpr int kFillPerSlot = (kFill + TW - 1) / TW;
  constexpr int SP = 32 * TW;                            // row pitch of the epilogue stash (floats)
  static_assert(TW >= 4, "the four A pieces of a unit are issued in the first four tile slots");
  half8 *lbuf = reinterpret_cast<half8 *>(gsm);          // [3][kStepVecs]: ring of k-step images (the allocation also covers the epilogue stash)
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t r = lane & 31u, g = lane >> 5;
  const uint32_t M = d.M, K = d.K, units = d.units;
  const uint64_t m0 = (uint64_t)blockIdx.x * 128u + wv * 32u;
  const uint64_t arow_i = min(m0 + r, (uint64_t)M - 1);  // rows past the end repeat the last row (never stored)
  const float *arow0 = d.A[0] + arow_i * d.lda[0] + 16 * g;
  const float *arow1 = (NBP == 2 || (MODE >= 2 && d.A[1])) ? d.A[1] + arow_i * d.lda[1] + 16 * g : arow0;
  const uint32_t gunits = NBP * units, steps = 2 * gunits;
  // Scale of this lane's row in a phase: from the row maxima the operand's producer left behind, or -- no array given: the
  // small batches, whose operands sit in the L2 and whose steps are bound by the host's launch rate -- from one more read
  // of the row (the two lanes that share a row hold its two 16-column halves of every unit).
  auto phase_scale = [&](int ph) -> float {
    if (d.aamax[ph]) return row_scale_of(d.aamax[ph][arow_i]);
    float mx = 0.f;
    for (uint32_t u = 0; u < units; u++) {
      const bool second = MODE >= 2 && NBP == 1 && u >= d.asplit;
      const float *base = (ph || second ? arow1 : arow0) + 32 * (second ? u - d.asplit : u);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (!kTail || 32 * u + 32 <= K) {
          mx = fmaxf(mx, amax4(*reinterpret_cast<const float4 *>(base + 4 * q)));
        } else {
#pragma unroll
          for (int c = 0; c < 4; c++) { const uint32_t k = 32 * u + 16 * g + 4 * q + c; if (k < K) mx = fmaxf(mx, fabsf(base[4 * q + c])); }
        }
      }
    }
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return row_scale_of(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
  };
  float asc = phase_scale(0);

  f32x16 acc[TW];
#pragma unroll
  for (int t = 0; t < TW; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

  float4 an[4];                                          // A of the next unit
  auto load_a_piece = [&](uint32_t gu, int q) {
    const bool ph = NBP == 2 && gu >= units;
    const uint32_t u = ph ? gu - units : gu;
    const bool second = MODE >= 2 && NBP == 1 && u >= d.asplit;      // (the K-concatenated operand's second tensor)
    const float *ptr = (ph || second ? arow1 : arow0) + 32 * (second ? u - d.asplit : u) + 4 * q;
    if (!kTail || 32 * u + 32 <= K) {
      an[q] = *reinterpret_cast<const float4 *>(ptr);
    } else {
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; c++) { const uint32_t k = 32 * u + 16 * g + 4 * q + c; v[c] = k < K ? ptr[c] : 0.f; }
      an[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto fill_b = [&](uint32_t s, int slot) {
    const half8 *src = d.Bimg + (size_t)s * kStepVecs;
    half8 *dst = lbuf + (size_t)(s % 3u) * kStepVecs;
#pragma unroll
    for (int q = slot * kFillPerSlot; q < (slot + 1) * kFillPerSlot && q < kFill; q++) {
      const uint32_t base = q * kThreads + wv * 64u;                     // wave-uniform
      if (base < (uint32_t)kStepVecs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + base + lane),
                                         (__attribute__((address_space(3))) void *)(dst + base), 16, 0, 0);
    }
  };

  // A 32 x 32 TW accumulator tile (C/D layout: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)) to global memory
  // through one ring slot: 8 rows x half the columns per wavefront at a time, leaving as float4s -- 512 contiguous bytes per
  // row and instruction (straight from the C/D layout it was 16 TW dword stores per lane: 8 % of the workgroup's lifetime).
  auto store_tile = [&](float *dst0, int64_t ld, uint32_t slot, uint32_t rows_ok, uint32_t cols_ok, int prod) {
    constexpr int GACT = MODE == 4 ? 1 : MODE == 5 ? 2 : -1;   // (GAT tail, MODE 3 / 4 / 5: the activation looked up / relu / elu)
    constexpr int CT = TW / 2;                            // column tiles per chunk: 8 rows x 32 CT floats per wavefront = 2 TW KB per workgroup (= one ring slot)
    float *zst = reinterpret_cast<float *>(lbuf + (size_t)slot * kStepVecs) + (size_t)wv * (8 * 32 * CT);
    const bool zvec = (ld & 3) == 0;
    // (GAT tail: a lane's column within a chunk is fixed -- 4 (lane % (8 CT)) -- so its four attention weights are too)
    // (GAT tail: a lane's column within a chunk is fixed -- 32 CT ch + 4 (lane % (8 CT)) -- so its four attention weights are
    //  too: loaded once, BEFORE the stores -- a load between them would wait for every store issued so far, vmcnt is shared)
    float *gat_u = MODE >= 3 ? d.gat_u[prod] : nullptr;
    float4 gat_a[2] = {f4zero(), f4zero()};
    const uint32_t gat_lmask = MODE >= 3 ? d.gat_ls - 1u : 0u, gat_hshift = MODE >= 3 ? (uint32_t)__builtin_ctz(d.gat_ls) + 2u : 0u;
    if constexpr (MODE >= 3) {
#pragma unroll
      for (int ch = 0; ch < 2; ch++) gat_a[ch] = ld4(d.gat_att[prod] + 32 * CT * ch + 4 * (lane % (8 * CT)));
    }
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int t = 0; t < CT; t++)
#pragma unroll
          for (int ii = 0; ii < 4; ii++) zst[(4 * g + ii) * (32 * CT) + 32 * t + r] = acc[CT * ch + t][4 * q + ii];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll(MODE >= 3 ? 1 : CT)
        for (int it = 0; it < CT; it++) {
          const uint32_t idx = it * 64 + lane, lr = idx / (8 * CT), c4 = idx % (8 * CT);
          const uint32_t rloc = 8 * q + lr, col = 32 * CT * ch + 4 * c4;
          if constexpr (MODE >= 3) {
            // (whole wavefront: the head-slice sums are DPP butterflies; N == 32 TW here, every column is inside)
            float4 v = *reinterpret_cast<const float4 *>(zst + lr * (32 * CT) + 4 * c4);
            const float4 hv = act4(GACT >= 0 ? GACT : d.gat_act, v);
            const float u = slice_sum(gat_dot4(gat_a[ch], hv), d.gat_ls);
            if (rloc < rows_ok) {
              if ((lane & gat_lmask) == 0) gat_u[(m0 + rloc) * d.gat_H + (col >> gat_hshift)] = u;
              if (d.gat_store_act[prod]) v = hv;
              *reinterpret_cast<float4 *>(dst0 + (m0 + rloc) * ld + col) = v;
            }
          } else if (rloc < rows_ok && col < cols_ok) {
            const float4 v = *reinterpret_cast<const float4 *>(zst + lr * (32 * CT) + 4 * c4);
            float *dst = dst0 + (m0 + rloc) * ld + col;
            if (zvec) *reinterpret_cast<float4 *>(dst) = v;
            else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  };

#pragma unroll
  for (int slot = 0; slot < TW; slot++) fill_b(0, slot);
#pragma unroll
  for (int slot = 0; slot < TW; slot++) fill_b(1, slot);                 // (steps >= 2 always)
#pragma unroll
  for (int q = 0; q < 4; q++) load_a_piece(0, q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float4 ac[4];
  for (uint32_t gu = 0; gu < gunits; gu++) {
#pragma unroll
    for (int q = 0; q < 4; q++) ac[q] = an[q];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t st = 2 * gu + h;
      const half8 *lb = lbuf + (size_t)(st % 3u) * kStepVecs;
      half8 ah, am;
      {
        const float x[8] = {ac[2 * h].x, ac[2 * h].y, ac[2 * h].z, ac[2 * h].w,
                            ac[2 * h + 1].x, ac[2 * h + 1].y, ac[2 * h + 1].z, ac[2 * h + 1].w};
        split8_f16(x, asc, ah, am);
      }
      // (pin: the A registers are consumed -- and waited for -- BEFORE this step issues new copies)
      __builtin_amdgcn_sched_barrier(0);
      half8 fb[2][2];
#pragma unroll
      for (int pc = 0; pc < 2; pc++) fb[0][pc] = lb[(pc * TW) * 64 + lane];
#pragma unroll
      for (int t = 0; t < TW; t++) {
        if (t + 1 < TW) {
#pragma unroll
          for (int pc = 0; pc < 2; pc++) fb[(t + 1) & 1][pc] = lb[(pc * TW + t + 1) * 64 + lane];
        }
        if (st + 2 < steps) fill_b(st + 2, t);              // two steps ahead: never waited for in this step
        if (h == 0 && t < 4 && gu + 1 < gunits) load_a_piece(gu + 1, t);
        const half8 bh = fb[t & 1][0], bm = fb[t & 1][1];
        // the two small terms first, the dominant product last
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // wait only for what is OLDER than this step's own copies, then a bare barrier
      if (gu + 1 < gunits) {
        if (h == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kFill + 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kFill) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                      // every wavefront is done with this step's buffer
    }
    if (NBP == 2 && gu + 1 == units) {
      // end of the first product: Z_0 leaves while the first images of the second product are already on their way; the
      // accumulators start over.  The ring slot of the step that has just finished is free (the copies in flight fill the
      // other two).
      // (the bounds are made opaque: as loop invariants the store predicates would be hoisted out of the k-loop and live
      //  in -- spilled -- scalar registers across it)
      uint32_t rows_ok = __builtin_amdgcn_readfirstlane((uint32_t)min((uint64_t)32, (uint64_t)M - min((uint64_t)M, m0))), cols_ok = d.N;
      asm volatile("" : "+s"(rows_ok), "+s"(cols_ok));
      unscale_tile<TW>(acc, 1.0f / asc, d.btrail[0] + 32 * TW, g, r);
      if (MODE >= 2 && d.cbias[0]) add_bias_tile<TW>(acc, d.cbias[0], d.N, r);
      asc = phase_scale(1);
      store_tile(d.Z[0], d.ldz[0], (2 * gu + 1) % 3u, rows_ok, cols_ok, 0);
#pragma unroll
      for (int t = 0; t < TW; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
      // the next step's copies (of step st + 3) go into this very slot: nobody may issue them while a slower wavefront
      // still reads its part of the stash (bare barrier: the copies in flight are not drained)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  if constexpr (MODE >= 2) {
    // plain product(s): the last tile leaves the same way (the ring is dead: every wavefront passed the last step's barrier)
    unscale_tile<TW>(acc, 1.0f / asc, d.btrail[NBP - 1] + 32 * TW, g, r);
    if (d.cbias[NBP - 1]) add_bias_tile<TW>(acc, d.cbias[NBP - 1], d.N, r);
    store_tile(d.Z[NBP - 1], d.ldz[NBP - 1], 0u, (uint32_t)min((uint64_t)32, (uint64_t)M - min((uint64_t)M, m0)), d.N, NBP - 1);
  } else {
  // ---- epilogue: the ring is dead (every wavefront passed the last step's barrier).  Specialised for the two activations
  //      the reference's configurations use (config_train: elu, relu) on full-width rows; everything else takes the generic
  //      copy (activation looked up per element, column predicates)
  unscale_tile<TW>(acc, 1.0f / asc, d.btrail[NBP - 1] + 32 * TW, g, r);
  const bool full = d.N == 32 * TW, same = NBA == 1 || d.act[1] == d.act[0];
  if ((MODE == 1 || MODE == -1) && d.stats_r && full && same && (d.act[0] == 1 || d.act[0] == 2)) {
    // (the row statistics saved by the forward epilogue: the specialised copies only -- what the benchmark's layers run)
    if (d.act[0] == 1) fused_epilogue<TW, MODE, NBA, 1, true, MODE == 1 || MODE == -1>(d, acc, gsm, m0);
    else fused_epilogue<TW, MODE, NBA, 2, true, MODE == 1 || MODE == -1>(d, acc, gsm, m0);
  }
  else if (full && same && d.act[0] == 1) fused_epilogue<TW, MODE, NBA, 1, true>(d, acc, gsm, m0);
  else if (full && same && d.act[0] == 2) fused_epilogue<TW, MODE, NBA, 2, true>(d, acc, gsm, m0);
  else fused_epilogue<TW, MODE, NBA, -1, false>(d, acc, gsm, m0);
  }
}

int fill_dropout(FusedDesc &p, float drop_p, uint64_t drop_seed, const char *who) {
  p.drop_thr = 0; p.drop_scale = 1.0f; p.seed_lo = (uint32_t)drop_seed; p.seed_hi = (uint32_t)(drop_seed >> 32);
  if (drop_p <= 0.f) return SG_OK;
  if (!(drop_p < 1.f)) return set_error(SG_ERR_INVALID, "%s: dropout probability %g", who, drop_p);
  p.drop_thr = drop_threshold16(drop_p);
  p.drop_scale = 1.0f / (1.0f - drop_p);
  return SG_OK;
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// layout of sl_gemm_act_norm_pack: nimg images back to back, then their trailers
inline void set_images(FusedDesc &p, const void *packed, int nimg, uint32_t N, uint32_t K) {
  const uint32_t tiles = N <= 128 ? 4u : 8u;
  const size_t ib = (size_t)((K + 31) / 32) * 4 * tiles * 64 * 16, tb = (size_t)2 * 32 * tiles * 4;
  const char *base = reinterpret_cast<const char *>(packed);
  p.Bimg = reinterpret_cast<const half8 *>(base);
  for (int b = 0; b < nimg; b++) p.btrail[b] = reinterpret_cast<const float *>(base + nimg * ib + b * tb);
}

template <int TW, int MODE, int NBP, int NBA>
int launch_fused(FusedDesc d, hipStream_t st) {
  // ring of three k-step images (3 x 2 TW KB) or the epilogue's stash (4 wavefronts x 16 rows x 32 TW floats), whichever is larger
  const size_t lds_ = MODE >= 2 ? (size_t)3 * 2 * TW * 64 * 16 : std::max<size_t>((size_t)3 * 2 * TW * 64 * 16, (size_t)4 * 16 * 32 * TW * 4);
  const size_t lds = lds_;
  const uint32_t grid = (d.M + 127) / 128;
  if (d.K % 32 == 0) {
    if (lds > 64 * 1024) SHD_HIP(ensure_dynamic_lds((const void *)gemm_nt_fused_kernel<TW, MODE, NBP, NBA, false>, lds));
    hipLaunchKernelGGL((gemm_nt_fused_kernel<TW, MODE, NBP, NBA, false>), dim3(grid), dim3(256), lds, st, d);
  } else {
    if (lds > 64 * 1024) SHD_HIP(ensure_dynamic_lds((const void *)gemm_nt_fused_kernel<TW, MODE, NBP, NBA, true>, lds));
    hipLaunchKernelGGL((gemm_nt_fused_kernel<TW, MODE, NBP, NBA, true>), dim3(grid), dim3(256), lds, st, d);
  }
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

bool g_fused_epilogue = true;      // (sl_set_fused_epilogue: the separate act + norm kernels instead, A/B and tests)

}  // namespace

// the reduction of the per-workgroup partial sums (aggregate.hip)
int act_norm_finish_launch(const float *partial, uint32_t nblocks, int nb, uint32_t F, float *dscale, float *doffset, float *dbias,
                           hipStream_t st);

}  // namespace shadow

using namespace shadow;

extern "C" int sl_set_fused_epilogue(int on) {
  const int prev = g_fused_epilogue ? 1 : 0;
  if (on >= 0) g_fused_epilogue = on != 0;
  return prev;
}

// column tiles of the B image the epilogue kernels stream: 4 or 8 (the image is zero-padded above N)
extern "C" uint32_t sl_gemm_act_norm_tiles(uint32_t N) { return N <= 128 ? 4u : 8u; }

// One weight image of the epilogue kernels: fp16 pieces in sl_gemm_act_norm_tiles(N) column tiles + the scale trailer.
extern "C" size_t sl_gemm_act_norm_pack_bytes(uint32_t N, uint32_t K) {
  const uint32_t tiles = sl_gemm_act_norm_tiles(N);
  return pack_f16_image_bytes(K, tiles) + pack_f16_trailer_bytes(tiles);
}

// The nb <= 2 weights W_b [N, K] of one launch: images back to back, then the trailers (nb x sl_gemm_act_norm_pack_bytes).
// d_zero / n_zero (may be NULL / 0): floats cleared by the same launch.
extern "C" int sl_gemm_act_norm_pack(int nb, const float *const *d_B, const int64_t *ldb, uint32_t N, uint32_t K, void *d_packed,
                                     float *d_zero, uint32_t n_zero, void *stream) {
  if (nb < 1 || nb > 2 || !d_B || !ldb || !d_packed) return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_pack: bad argument");
  const uint32_t tiles = sl_gemm_act_norm_tiles(N);
  const size_t ib = pack_f16_image_bytes(K, tiles), tb = pack_f16_trailer_bytes(tiles);
  char *base = reinterpret_cast<char *>(d_packed);
  PackF16Src src[2];
  for (int b = 0; b < nb; b++)
    src[b] = PackF16Src{d_B[b], d_B[b], ldb[b], 1, ldb[b], 1, K, base + b * ib, reinterpret_cast<float *>(base + nb * ib + b * tb)};
  return pack_f16(nb, src, N, K, tiles, d_zero, n_zero, (hipStream_t)stream);
}

// The general form for one image: B element (j, k) = B1[j s1j + k s1k] for k < K1, B2[j s2j + (k - K1) s2k] behind
// (a transposed weight, [Ws^T | Wn^T] without materialising the concatenation) -- the operand of sl_gemm_an_bwd.
extern "C" int sl_gemm_act_norm_pack_b2(const float *d_B1, int64_t s1j, int64_t s1k, uint32_t K1, const float *d_B2, int64_t s2j,
                                        int64_t s2k, uint32_t N, uint32_t K, void *d_packed, void *stream) {
  if (!d_packed) return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_pack_b2: null argument");
  const uint32_t tiles = sl_gemm_act_norm_tiles(N);
  char *base = reinterpret_cast<char *>(d_packed);
  const PackF16Src src{d_B1, d_B2, s1j, s1k, s2j, s2k, K1, base, reinterpret_cast<float *>(base + pack_f16_image_bytes(K, tiles))};
  return pack_f16(1, &src, N, K, tiles, nullptr, 0, (hipStream_t)stream);
}

extern "C" int sl_gemm_act_norm_supported(uint32_t N, uint32_t K) {
  return g_fused_epilogue && N >= 16 && N <= 256 && (N & 3) == 0 && K > 0;
}

extern "C" int sl_gemm_act_norm_fwd(int nb, const float *const *d_A, const int64_t *lda, const float *const *d_a_amax,
                                    const void *d_packed_B, uint32_t M,
                                    uint32_t N, uint32_t K, float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                                    const int *act, const float *d_scale, const float *d_offset, float out_scale, float *d_out,
                                    int64_t ldo, float drop_p, uint64_t drop_seed, float *d_out_dropped, int64_t ldo_dropped,
                                    float *d_out_amax, float *d_row_stats, void *stream) {
  if (nb < 1 || nb > 2) return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: nb must be 1 or 2");
  if (!d_A || !lda || !d_packed_B || !d_Z || !ldz || !act || !d_scale || !d_offset || !d_out)
    return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: null argument");
  if (M == 0) return SG_OK;
  if (N < 16 || N > 256 || (N & 3) || K == 0)
    return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: N = %u, K = %u (N a multiple of 4 in [16, 256])", N, K);
  FusedDesc p;
  memset(&p, 0, sizeof(p));
  for (int b = 0; b < nb; b++) {
    if (!d_A[b] || !d_Z[b]) return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: null operand of branch %d", b);
    if ((lda[b] & 3) || !al16(d_A[b]) || (ldz[b] & 3) || !al16(d_Z[b]))
      return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: operands must be 16-byte aligned with ld %% 4 == 0");
    if (act[b] < 0 || act[b] > 4) return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: unknown activation %d", act[b]);
    if (d_bias && d_bias[b] && !al16(d_bias[b])) return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: bias must be 16-byte aligned");
    p.A[b] = d_A[b]; p.lda[b] = lda[b]; p.Z[b] = d_Z[b]; p.ldz[b] = ldz[b]; p.act[b] = act[b]; p.aamax[b] = d_a_amax ? d_a_amax[b] : nullptr;
    p.bias[b] = d_bias ? d_bias[b] : nullptr;
  }
  set_images(p, d_packed_B, nb, N, K);
  if (!al16(d_scale) || !al16(d_offset) || !al16(d_out) || (ldo & 3))
    return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: scale / offset / out must be 16-byte aligned, ldo %% 4 == 0");
  p.M = M; p.N = N; p.K = K; p.units = (K + 31) / 32; p.asplit = p.units;
  p.scale = d_scale; p.offset = d_offset; p.out_scale = out_scale; p.eps = 1e-9f;
  p.out = d_out; p.ldo = ldo; p.out_amax = d_out_amax; p.stats_w = d_row_stats;
  int rc;
  if ((rc = fill_dropout(p, drop_p, drop_seed, "sl_gemm_act_norm_fwd")) != SG_OK) return rc;
  if (d_out_dropped) {
    if (!p.drop_thr) return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: a dropped output needs drop_p > 0");
    if ((ldo_dropped & 3) || !al16(d_out_dropped))
      return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: the dropped output must be 16-byte aligned, ld %% 4 == 0");
    p.out2 = d_out_dropped; p.ldo2 = ldo_dropped;
  }
  hipStream_t st = (hipStream_t)stream;
  if (N <= 128) return nb == 1 ? launch_fused<4, 0, 1, 1>(p, st) : launch_fused<4, 0, 2, 2>(p, st);
  return nb == 1 ? launch_fused<8, 0, 1, 1>(p, st) : launch_fused<8, 0, 2, 2>(p, st);
}

// Plain products on the same kernel (no epilogue arithmetic beyond an optional bias): C_b = A_b . W_b^T + bias_b for
// b < nb <= 2 in one launch.
extern "C" int sl_gemm_nt2_f32(int nb, const float *const *d_A, const int64_t *lda, const float *const *d_a_amax, const void *d_packed_B,
                               uint32_t M, uint32_t N, uint32_t K, const float *const *d_bias, float *const *d_C, const int64_t *ldc,
                               void *stream) {
  if (nb < 1 || nb > 2 || !d_A || !lda || !d_packed_B || !d_C || !ldc) return set_error(SG_ERR_INVALID, "sl_gemm_nt2_f32: bad argument");
  if (M == 0) return SG_OK;
  if (N < 16 || N > 256 || (N & 3) || K == 0)
    return set_error(SG_ERR_INVALID, "sl_gemm_nt2_f32: N = %u, K = %u (N a multiple of 4 in [16, 256])", N, K);
  FusedDesc p;
  memset(&p, 0, sizeof(p));
  for (int b = 0; b < nb; b++) {
    if (!d_A[b] || !d_C[b]) return set_error(SG_ERR_INVALID, "sl_gemm_nt2_f32: null operand of product %d", b);
    if ((lda[b] & 3) || !al16(d_A[b]) || (ldc[b] & 3) || !al16(d_C[b]))
      return set_error(SG_ERR_INVALID, "sl_gemm_nt2_f32: operands must be 16-byte aligned with ld %% 4 == 0");
    p.A[b] = d_A[b]; p.lda[b] = lda[b]; p.Z[b] = d_C[b]; p.ldz[b] = ldc[b]; p.aamax[b] = d_a_amax ? d_a_amax[b] : nullptr;
    p.cbias[b] = d_bias ? d_bias[b] : nullptr;
  }
  set_images(p, d_packed_B, nb, N, K);
  p.M = M; p.N = N; p.K = K; p.units = (K + 31) / 32; p.asplit = p.units;
  hipStream_t st = (hipStream_t)stream;
  if (N <= 128) return nb == 1 ? launch_fused<4, 2, 1, 1>(p, st) : launch_fused<4, 2, 2, 1>(p, st);
  return nb == 1 ? launch_fused<8, 2, 1, 1>(p, st) : launch_fused<8, 2, 2, 1>(p, st);
}

// The GAT layer's two Linears of one input (shaDow/layers.py:604-611) WITH the per-node attention terms of layers.py:566-569:
//   C0 = A W0^T + b0 = z_self,  u_s[row, h] = att[0, h, :] . act(z_self[row, head h]);
//   C1 = act(A W1^T + b1) = hn (z_neigh itself is not written: every later use needs act(z_neigh) or, in the backward pass, a
//   derivative that follows from it -- gat_act.h),  u_n[row, h] = att[1, h, :] . hn[row, head h]
// computed where the tiles leave the kernel: gat_node_fwd_kernel's pass over z_self / z_neigh (0.18 ms per layer at 294 k rows)
// is not run, bit-identical u_s / u_n / hn (same activation, same dot4 + butterfly per head slice).
// N = heads * D == 256 with D / 4 a power of two <= 32; otherwise SG_ERR_INVALID (callers keep sl_gemm_nt2_f32 + sl_gat_fwd).
extern "C" int sl_gemm_nt2_gat_supported(uint32_t N, uint32_t heads) {
  if (N != 256 || heads == 0 || N % heads) return 0;
  const uint32_t ls = N / heads / 4;
  return (N / heads) % 4 == 0 && ls >= 1 && ls <= 32 && (ls & (ls - 1)) == 0;
}

extern "C" int sl_gemm_nt2_gat_f32(const float *d_A, int64_t lda, const float *d_a_amax, const void *d_packed_B, uint32_t M, uint32_t N,
                                   uint32_t K, const float *const *d_bias, float *d_z_self, int64_t ldzs, float *d_hn, int64_t ldhn,
                                   const float *d_att, int act, uint32_t heads, float *d_u_s, float *d_u_n, void *stream) {
  if (!d_A || !d_packed_B || !d_z_self || !d_hn || !d_att || !d_u_s || !d_u_n) return set_error(SG_ERR_INVALID, "sl_gemm_nt2_gat_f32: null argument");
  if (M == 0) return SG_OK;
  if (!sl_gemm_nt2_gat_supported(N, heads) || K == 0 || act < 0 || act > 4)
    return set_error(SG_ERR_INVALID, "sl_gemm_nt2_gat_f32: N = %u, heads = %u, act = %d (N == 256, head width 4 * 2^k <= 128)", N, heads, act);
  if ((lda & 3) || !al16(d_A) || (ldzs & 3) || !al16(d_z_self) || (ldhn & 3) || !al16(d_hn) || !al16(d_att))
    return set_error(SG_ERR_INVALID, "sl_gemm_nt2_gat_f32: operands must be 16-byte aligned with ld %% 4 == 0");
  FusedDesc p;
  memset(&p, 0, sizeof(p));
  for (int b = 0; b < 2; b++) {
    p.A[b] = d_A; p.lda[b] = lda; p.aamax[b] = d_a_amax; p.cbias[b] = d_bias ? d_bias[b] : nullptr;
    p.gat_att[b] = d_att + (size_t)b * N;
  }
  p.Z[0] = d_z_self; p.ldz[0] = ldzs; p.Z[1] = d_hn; p.ldz[1] = ldhn;
  p.gat_u[0] = d_u_s; p.gat_u[1] = d_u_n; p.gat_store_act[0] = 0; p.gat_store_act[1] = 1;
  p.gat_act = act; p.gat_ls = N / heads / 4; p.gat_H = heads;
  set_images(p, d_packed_B, 2, N, K);
  p.M = M; p.N = N; p.K = K; p.units = (K + 31) / 32; p.asplit = p.units;
  hipStream_t st = (hipStream_t)stream;
  return act == 2 ? launch_fused<8, 5, 2, 1>(p, st) : act == 1 ? launch_fused<8, 4, 2, 1>(p, st) : launch_fused<8, 3, 2, 1>(p, st);
}

// C = [A0 | A1] . B^T with the K-concatenated operand in two tensors (K0 columns from A0, K - K0 from A1; K0 % 32 == 0);
// d_packed_B = sl_gemm_act_norm_pack_b2 of the matching concatenated weight.  d_a_amax: max over BOTH parts of a row (or NULL).
extern "C" int sl_gemm_nt_cat_f32(const float *d_A0, int64_t lda0, uint32_t K0, const float *d_A1, int64_t lda1, const float *d_a_amax,
                                  const void *d_packed_B, uint32_t M, uint32_t N, uint32_t K, const float *d_bias, float *d_C,
                                  int64_t ldc, void *stream) {
  if (!d_A0 || !d_A1 || !d_packed_B || !d_C) return set_error(SG_ERR_INVALID, "sl_gemm_nt_cat_f32: null argument");
  if (M == 0) return SG_OK;
  if (N < 16 || N > 256 || (N & 3) || K0 == 0 || K0 >= K || (K0 & 31))
    return set_error(SG_ERR_INVALID, "sl_gemm_nt_cat_f32: N = %u, K0 = %u of K = %u (N %% 4 == 0 in [16, 256], K0 %% 32 == 0)", N, K0, K);
  if ((lda0 & 3) || !al16(d_A0) || (lda1 & 3) || !al16(d_A1) || (ldc & 3) || !al16(d_C))
    return set_error(SG_ERR_INVALID, "sl_gemm_nt_cat_f32: operands must be 16-byte aligned with ld %% 4 == 0");
  FusedDesc p;
  memset(&p, 0, sizeof(p));
  p.A[0] = d_A0; p.lda[0] = lda0; p.A[1] = d_A1; p.lda[1] = lda1; p.Z[0] = d_C; p.ldz[0] = ldc; p.aamax[0] = d_a_amax; p.cbias[0] = d_bias;
  set_images(p, d_packed_B, 1, N, K);
  p.M = M; p.N = N; p.K = K; p.units = (K + 31) / 32; p.asplit = K0 / 32;
  hipStream_t st = (hipStream_t)stream;
  return N <= 128 ? launch_fused<4, 2, 1, 1>(p, st) : launch_fused<8, 2, 1, 1>(p, st);
}

extern "C" size_t sl_gemm_an_bwd_partial_floats(uint32_t M, uint32_t N, int nb) {
  return (size_t)((M + 127) / 128) * (size_t)nb * 3 * N;
}

extern "C" int sl_gemm_an_bwd(const float *d_A, int64_t lda, const float *d_a_amax, const void *d_packed_B, uint32_t M, uint32_t N,
                              uint32_t K, int nb,
                              const float *const *d_Z, const int64_t *ldz, const float *const *d_bias, const int *act,
                              const float *d_scale, const float *d_offset, float out_scale, float *const *d_dZ,
                              const int64_t *lddz, float *d_dscale, float *d_doffset, float *d_dbias, float *d_partial,
                              float drop_p, uint64_t drop_seed, float *d_dz0_amax, const float *d_row_stats, void *stream) {
  return sl_gemm_an_bwd_corr(d_A, lda, d_a_amax, d_packed_B, M, N, K, nb, d_Z, ldz, d_bias, act, d_scale, d_offset, out_scale, d_dZ, lddz,
                             d_dscale, d_doffset, d_dbias, d_partial, drop_p, drop_seed, d_dz0_amax, d_row_stats, nullptr, 0, nullptr, 0, stream);
}

extern "C" int sl_gemm_an_bwd_corr(const float *d_A, int64_t lda, const float *d_a_amax, const void *d_packed_B, uint32_t M, uint32_t N,
                                   uint32_t K, int nb,
                                   const float *const *d_Z, const int64_t *ldz, const float *const *d_bias, const int *act,
                                   const float *d_scale, const float *d_offset, float out_scale, float *const *d_dZ,
                                   const int64_t *lddz, float *d_dscale, float *d_doffset, float *d_dbias, float *d_partial,
                                   float drop_p, uint64_t drop_seed, float *d_dz0_amax, const float *d_row_stats,
                                   const float *d_corr, int64_t ldcorr, const uint32_t *d_corr_row, uint32_t corr_rows, void *stream) {
  return sl_gemm_an_bwd_plain(d_A, lda, d_a_amax, d_packed_B, M, N, K, nb, d_Z, ldz, d_bias, act, d_scale, d_offset, out_scale, d_dZ, lddz,
                              d_dscale, d_doffset, d_dbias, d_partial, drop_p, drop_seed, d_dz0_amax, d_row_stats, d_corr, ldcorr, d_corr_row,
                              corr_rows, nullptr, 0, nullptr, stream);
}

// ... and with a DENSE addend d_dout_plain [M, N] (pitch lddp) that does not pass the dropout mask: the layer below is in
// dual-output mode (FusedDesc::dplain).  N == 256 only (the benchmark width's instantiation).  d_plain_row: the addend of row i is
// row d_plain_row[i] of d_dout_plain (a pooled read-out's gradient table, sl_pool_grad_table).
extern "C" int sl_gemm_an_bwd_plain(const float *d_A, int64_t lda, const float *d_a_amax, const void *d_packed_B, uint32_t M, uint32_t N,
                                    uint32_t K, int nb,
                                    const float *const *d_Z, const int64_t *ldz, const float *const *d_bias, const int *act,
                                    const float *d_scale, const float *d_offset, float out_scale, float *const *d_dZ,
                                    const int64_t *lddz, float *d_dscale, float *d_doffset, float *d_dbias, float *d_partial,
                                    float drop_p, uint64_t drop_seed, float *d_dz0_amax, const float *d_row_stats,
                                    const float *d_corr, int64_t ldcorr, const uint32_t *d_corr_row, uint32_t corr_rows,
                                    const float *d_dout_plain, int64_t lddp, const uint32_t *d_plain_row, void *stream) {
  if (d_plain_row && !d_dout_plain) return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd_plain: a row map without the addend's table");
  if (d_dout_plain && (N <= 128 || (lddp & 3) || !al16(d_dout_plain) || lddp < (int64_t)N))
    return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd_plain: the dense addend needs 128 < N <= 256, ld %% 4 == 0 >= N and 16-byte alignment");
  if (d_corr && (!d_corr_row || (ldcorr & 3) || !al16(d_corr) || ldcorr < (int64_t)N))
    return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd_corr: the sparse addend needs its row map, ld %% 4 == 0 >= N and 16-byte alignment");
  if (nb != 2) return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd: nb must be 2 (a GraphSAGE layer below)");
  if (!d_A || !d_packed_B || !d_Z || !ldz || !act || !d_scale || !d_offset || !d_dZ || !lddz || !d_dscale || !d_doffset || !d_partial)
    return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    SHD_HIP(hipMemsetAsync(d_dscale, 0, (size_t)nb * N * 4, st));
    SHD_HIP(hipMemsetAsync(d_doffset, 0, (size_t)nb * N * 4, st));
    if (d_dbias) SHD_HIP(hipMemsetAsync(d_dbias, 0, (size_t)nb * N * 4, st));
    return SG_OK;
  }
  if (N < 16 || N > 256 || (N & 3) || K == 0 || (lda & 3) || !al16(d_A))
    return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd: N = %u, K = %u, lda = %lld unsupported", N, K, (long long)lda);
  FusedDesc p;
  memset(&p, 0, sizeof(p));
  p.A[0] = d_A; p.lda[0] = lda; p.aamax[0] = d_a_amax;
  set_images(p, d_packed_B, 1, N, K);
  for (int b = 0; b < nb; b++) {
    if (!d_Z[b] || !d_dZ[b]) return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd: null operand of branch %d", b);
    if ((ldz[b] & 3) || !al16(d_Z[b]) || (lddz[b] & 3) || !al16(d_dZ[b]))
      return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd: operands must be 16-byte aligned with ld %% 4 == 0");
    if (act[b] < 0 || act[b] > 4) return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd: unknown activation %d", act[b]);
    if (d_bias && d_bias[b] && !al16(d_bias[b])) return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd: bias must be 16-byte aligned");
    p.Zr[b] = d_Z[b]; p.ldzr[b] = ldz[b]; p.dZ[b] = d_dZ[b]; p.lddz[b] = lddz[b]; p.act[b] = act[b];
    p.bias[b] = d_bias ? d_bias[b] : nullptr;
  }
  if (!al16(d_scale) || !al16(d_offset)) return set_error(SG_ERR_INVALID, "sl_gemm_an_bwd: scale / offset must be 16-byte aligned");
  p.M = M; p.N = N; p.K = K; p.units = (K + 31) / 32; p.asplit = p.units;
  p.scale = d_scale; p.offset = d_offset; p.out_scale = out_scale; p.eps = 1e-9f;
  p.partial = d_partial; p.dz_amax = d_dz0_amax;
  p.stats_r = d_row_stats;
  p.corr = d_corr; p.ldcorr = ldcorr; p.corr_row = d_corr_row; p.corr_rows = d_corr ? corr_rows : 0;
  p.dplain = d_dout_plain; p.lddplain = lddp; p.plain_row = d_plain_row;
  int rc;
  if ((rc = fill_dropout(p, drop_p, drop_seed, "sl_gemm_an_bwd")) != SG_OK) return rc;
  rc = d_dout_plain ? launch_fused<8, -1, 1, 2>(p, st) : (N <= 128 ? launch_fused<4, 1, 1, 2>(p, st) : launch_fused<8, 1, 1, 2>(p, st));
  if (rc != SG_OK) return rc;
  return act_norm_finish_launch(d_partial, (M + 127) / 128, nb, N, d_dscale, d_doffset, d_dbias, st);
}
