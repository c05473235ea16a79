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
// there 16 rows at a time (C/D layout -> row-major, conflict-free ds_write_b32) and then walks the rows ONE ROW PER
// WAVEFRONT with a float4 per lane -- the layout and the arithmetic of act_norm_kernel<64, 64, ...> (aggregate.hip), so
// every global access of the epilogue is a whole 1-KiB row and the row statistics are plain 64-lane butterfly sums.
// The other workgroup resident on the CU keeps the matrix cores busy meanwhile (two 4-wave workgroups per CU).
#include <string.h>

#include <algorithm>

#include "actnorm_common.h"
#include "common.h"
#include "gemm_common.h"

namespace shadow {
namespace {

struct FusedDesc {
  // GEMM: C[M, N] = A_p[M, K] . B_p[N, K]^T for the phases p < NBP; the B images lie back to back
  const float *A[2];
  int64_t lda[2];
  const bf16x8 *Bimg;
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
  // backward: the layer below
  const float *Zr[2]; int64_t ldzr[2];
  float *dZ[2];       int64_t lddz[2];
  float *partial;                   // [grid, nb, 3, N]
};

__device__ __forceinline__ float row_sum64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// out row = out_scale-less sum_b norm_b(act(z_b + bias_b)) for the 4 columns f .. f + 3 of one row (all 64 lanes = one row)
template <int NBA>
__device__ __forceinline__ float4 an_row_fwd(const FusedDesc &d, uint32_t f, bool lane_on, const float4 (&zc)[NBA], float inv_seg) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int b = 0; b < NBA; b++) {
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f), h = z;
    if (lane_on) {
      z = zc[b];
      if (d.bias[b]) { const float4 bb = ld4(d.bias[b] + f); z.x += bb.x; z.y += bb.y; z.z += bb.z; z.w += bb.w; }
      h = make_float4(act_fwd(d.act[b], z.x), act_fwd(d.act[b], z.y), act_fwd(d.act[b], z.z), act_fwd(d.act[b], z.w));
    }
    // biased mean / variance over the row (layers.py:334-335)
    const float mean = row_sum64(h.x + h.y + h.z + h.w) * inv_seg;
    float4 dd = make_float4(h.x - mean, h.y - mean, h.z - mean, h.w - mean);
    if (!lane_on) dd = make_float4(0.f, 0.f, 0.f, 0.f);
    const float var = row_sum64(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z + dd.w * dd.w) * inv_seg + d.eps;
    const float rstd = rsqrtf(var);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), of = sc;
    if (lane_on) { sc = ld4(d.scale + (size_t)b * d.N + f); of = ld4(d.offset + (size_t)b * d.N + f); }
    // (x - mean) * scale * rsqrt(var) + offset   (layers.py:336)
    acc.x += dd.x * sc.x * rstd + of.x; acc.y += dd.y * sc.y * rstd + of.y;
    acc.z += dd.z * sc.z * rstd + of.z; acc.w += dd.w * sc.w * rstd + of.w;
  }
  return acc;
}

// act_norm backward of one row: dy = gradient of the row's (possibly dropped) output; writes dZ_b, accumulates the
// column sums of dscale (gs), doffset (go) and dbias (gb)
template <int NBA>
__device__ __forceinline__ void an_row_bwd(const FusedDesc &d, uint64_t r, uint32_t f, bool lane_on, float4 dy, const float4 (&zc)[NBA],
                                           float inv_seg, float4 (&gs)[NBA], float4 &go, float4 (&gb)[NBA]) {
  float4 dm = make_float4(d.out_scale, d.out_scale, d.out_scale, d.out_scale);
  if (d.drop_thr) {            // gradient of the fused output dropout: same mask, same 1 / (1 - p)
    const uint32_t keep = drop_keep4_raw(d.seed_lo, d.seed_hi, d.drop_thr, r, f);
    const float ks = d.out_scale * d.drop_scale;
    dm = make_float4((keep & 1u) ? ks : 0.f, (keep & 2u) ? ks : 0.f, (keep & 4u) ? ks : 0.f, (keep & 8u) ? ks : 0.f);
  }
  dy.x *= dm.x; dy.y *= dm.y; dy.z *= dm.z; dy.w *= dm.w;
#pragma unroll
  for (int b = 0; b < NBA; b++) {
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f), h = z;
    if (lane_on) {
      z = zc[b];
      if (d.bias[b]) { const float4 bb = ld4(d.bias[b] + f); z.x += bb.x; z.y += bb.y; z.z += bb.z; z.w += bb.w; }
      h = make_float4(act_fwd(d.act[b], z.x), act_fwd(d.act[b], z.y), act_fwd(d.act[b], z.z), act_fwd(d.act[b], z.w));
    }
    const float mean = row_sum64(h.x + h.y + h.z + h.w) * inv_seg;
    float4 dd = make_float4(h.x - mean, h.y - mean, h.z - mean, h.w - mean);
    if (!lane_on) dd = make_float4(0.f, 0.f, 0.f, 0.f);
    const float var = row_sum64(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z + dd.w * dd.w) * inv_seg + d.eps;
    const float rstd = rsqrtf(var);
    const float4 xh = make_float4(dd.x * rstd, dd.y * rstd, dd.z * rstd, dd.w * rstd);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane_on) sc = ld4(d.scale + (size_t)b * d.N + f);
    gs[b].x += dy.x * xh.x; gs[b].y += dy.y * xh.y; gs[b].z += dy.z * xh.z; gs[b].w += dy.w * xh.w;
    if (b == 0) { go.x += dy.x; go.y += dy.y; go.z += dy.z; go.w += dy.w; }
    const float4 dxh = make_float4(dy.x * sc.x, dy.y * sc.y, dy.z * sc.z, dy.w * sc.w);
    const float m1 = row_sum64(dxh.x + dxh.y + dxh.z + dxh.w) * inv_seg;
    const float m2 = row_sum64(dxh.x * xh.x + dxh.y * xh.y + dxh.z * xh.z + dxh.w * xh.w) * inv_seg;
    float4 dh = make_float4(rstd * (dxh.x - m1 - xh.x * m2), rstd * (dxh.y - m1 - xh.y * m2),
                            rstd * (dxh.z - m1 - xh.z * m2), rstd * (dxh.w - m1 - xh.w * m2));
    if (lane_on) {
      dh.x *= act_bwd(d.act[b], z.x, h.x); dh.y *= act_bwd(d.act[b], z.y, h.y);
      dh.z *= act_bwd(d.act[b], z.z, h.z); dh.w *= act_bwd(d.act[b], z.w, h.w);
      st4s(d.dZ[b] + (int64_t)r * d.lddz[b] + f, dh);
      gb[b].x += dh.x; gb[b].y += dh.y; gb[b].z += dh.z; gb[b].w += dh.w;
    }
  }
}

// TW column tiles (N <= 32 TW); MODE 0 forward / 1 backward; NBP GEMM phases; NBA act_norm branches;
// kTail: K % 32 != 0 (the last unit of a phase is zero-padded).  Main loop = gemm_nt_split_kernel<1, TW, 1, 4>.
template <int TW, int MODE, int NBP, int NBA, bool kTail>
__global__ void __launch_bounds__(256, 2) gemm_nt_fused_kernel(const FusedDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  constexpr int kThreads = 256;
  constexpr int kStepVecs = 3 * TW * 64;                 // bf16x8 vectors of one k-step's B image
  constexpr int kFill = (kStepVecs + kThreads - 1) / kThreads;
  constexpr int kFillPerSlot = (kFill + TW - 1) / TW;
  constexpr int SP = 32 * TW;                            // row pitch of the epilogue stash (floats)
  static_assert(TW >= 4, "the four A pieces of a unit are issued in the first four tile slots");
  static_assert(4 * 16 * SP * 4 <= 3 * kStepVecs * 16, "the stash must fit the dead B ring");
  bf16x8 *lbuf = reinterpret_cast<bf16x8 *>(gsm);        // [3][kStepVecs]: ring of k-step images
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t r = lane & 31u, g = lane >> 5;
  const uint32_t M = d.M, K = d.K, units = d.units;
  const uint64_t m0 = (uint64_t)blockIdx.x * 128u + wv * 32u;
  const uint64_t arow_i = min(m0 + r, (uint64_t)M - 1);  // rows past the end repeat the last row (never stored)
  const float *arow0 = d.A[0] + arow_i * d.lda[0] + 16 * g;
  const float *arow1 = NBP == 2 ? d.A[1] + arow_i * d.lda[1] + 16 * g : arow0;
  const uint32_t gunits = NBP * units, steps = 2 * gunits;

  f32x16 acc[TW];
#pragma unroll
  for (int t = 0; t < TW; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

  float4 an[4];                                          // A of the next unit
  auto load_a_piece = [&](uint32_t gu, int q) {
    const bool ph = NBP == 2 && gu >= units;
    const uint32_t u = ph ? gu - units : gu;
    const float *ptr = (ph ? arow1 : arow0) + 32 * u + 4 * q;
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
    const bf16x8 *src = d.Bimg + (size_t)s * kStepVecs;
    bf16x8 *dst = lbuf + (size_t)(s % 3u) * kStepVecs;
#pragma unroll
    for (int q = slot * kFillPerSlot; q < (slot + 1) * kFillPerSlot && q < kFill; q++) {
      const uint32_t base = q * kThreads + wv * 64u;                     // wave-uniform
      if (base < (uint32_t)kStepVecs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + base + lane),
                                         (__attribute__((address_space(3))) void *)(dst + base), 16, 0, 0);
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
      const bf16x8 *lb = lbuf + (size_t)(st % 3u) * kStepVecs;
      bf16x8 ah, am, al;
      {
        const float x[8] = {ac[2 * h].x, ac[2 * h].y, ac[2 * h].z, ac[2 * h].w,
                            ac[2 * h + 1].x, ac[2 * h + 1].y, ac[2 * h + 1].z, ac[2 * h + 1].w};
        split8(x, ah, am, al);
      }
      // (pin: the A registers are consumed -- and waited for -- BEFORE this step issues new copies)
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 fb[2][3];
#pragma unroll
      for (int pc = 0; pc < 3; pc++) fb[0][pc] = lb[(pc * TW) * 64 + lane];
#pragma unroll
      for (int t = 0; t < TW; t++) {
        if (t + 1 < TW) {
#pragma unroll
          for (int pc = 0; pc < 3; pc++) fb[(t + 1) & 1][pc] = lb[(pc * TW + t + 1) * 64 + lane];
        }
        if (st + 2 < steps) fill_b(st + 2, t);              // two steps ahead: never waited for in this step
        if (h == 0 && t < 4 && gu + 1 < gunits) load_a_piece(gu + 1, t);
        const bf16x8 bh = fb[t & 1][0], bm = fb[t & 1][1], bl = fb[t & 1][2];
        // small terms first, the dominant product last
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
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
      // end of the first product: Z_0 leaves in the C/D layout (col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5))
      // while the first images of the second product are already on their way; the accumulators start over
      // (the bounds are made opaque here: as loop invariants the 128 store predicates would be hoisted out of the k-loop
      //  and live in -- spilled -- scalar registers across it)
      uint32_t rows_ok = __builtin_amdgcn_readfirstlane((uint32_t)min((uint64_t)32, (uint64_t)M - min((uint64_t)M, m0))), cols_ok = d.N;
      asm volatile("" : "+s"(rows_ok), "+s"(cols_ok));
      float *zrow = d.Z[0] + (m0 + 4 * g) * d.ldz[0] + r;
      const uint32_t rlim = rows_ok > 4 * g ? rows_ok - 4 * g : 0u;       // rows (i & 3) + 8 (i >> 2) below this are stored
#pragma unroll
      for (int t = 0; t < TW; t++) {
        const bool cok = 32 * t + r < cols_ok;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          if (cok && (uint32_t)((i & 3) + 8 * (i >> 2)) < rlim) zrow[((i & 3) + 8 * (i >> 2)) * d.ldz[0] + 32 * t] = acc[t][i];
        }
      }
#pragma unroll
      for (int t = 0; t < TW; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    }
  }

  // ---- epilogue: the ring is dead (every wavefront passed the last step's barrier)
  float *stash = reinterpret_cast<float *>(gsm) + (size_t)wv * (16 * SP);
  const uint32_t f = 4 * lane;
  const bool lane_on = f < d.N;
  const float inv_seg = 1.0f / (float)d.N;
  float4 gs[NBA], go, gb[NBA];
  go = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int b = 0; b < NBA; b++) gs[b] = gb[b] = go;
  if (MODE == 0 && NBA == 2) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // Z_0 is re-read below by other lanes
#pragma unroll
  for (int hf = 0; hf < 2; hf++) {
    // rows 16 hf .. 16 hf + 15 of the tile: i = 8 hf + ii -> local row (ii & 3) + 8 (ii >> 2) + 4 g
#pragma unroll
    for (int t = 0; t < TW; t++)
#pragma unroll
      for (int ii = 0; ii < 8; ii++) stash[((ii & 3) + 8 * (ii >> 2) + 4 * g) * SP + 32 * t + r] = acc[t][8 * hf + ii];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint64_t rbase = m0 + 16u * hf;
    if (MODE == 0) {
      float4 zpre = make_float4(0.f, 0.f, 0.f, 0.f);
      if (NBA == 2 && rbase < M && lane_on) zpre = ld4(d.Z[0] + rbase * d.ldz[0] + f);
      for (int lr = 0; lr < 16; lr++) {
        const uint64_t row = rbase + lr;
        if (row >= M) break;                              // wave-uniform
        float4 zc[NBA];
        zc[NBA - 1] = lane_on ? *reinterpret_cast<const float4 *>(stash + lr * SP + f) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (NBA == 2) {
          zc[0] = zpre;
          if (lr + 1 < 16 && row + 1 < M && lane_on) zpre = ld4(d.Z[0] + (row + 1) * d.ldz[0] + f);
        }
        if (lane_on) st4s(d.Z[NBA - 1] + row * d.ldz[NBA - 1] + f, zc[NBA - 1]);
        float4 o = an_row_fwd<NBA>(d, f, lane_on, zc, inv_seg);
        if (lane_on) {
          o.x *= d.out_scale; o.y *= d.out_scale; o.z *= d.out_scale; o.w *= d.out_scale;
          if (d.drop_thr) {
            const uint32_t keep = drop_keep4_raw(d.seed_lo, d.seed_hi, d.drop_thr, row, f);
            const float4 dr = make_float4((keep & 1u) ? o.x * d.drop_scale : 0.f, (keep & 2u) ? o.y * d.drop_scale : 0.f,
                                          (keep & 4u) ? o.z * d.drop_scale : 0.f, (keep & 8u) ? o.w * d.drop_scale : 0.f);
            if (d.out2) st4s(d.out2 + row * d.ldo2 + f, dr);      // dual mode: out stays un-dropped
            else o = dr;
          }
          st4s(d.out + row * d.ldo + f, o);
        }
      }
    } else {
      float4 zpre[NBA];
#pragma unroll
      for (int b = 0; b < NBA; b++)
        zpre[b] = (rbase < M && lane_on) ? ld4s(d.Zr[b] + rbase * d.ldzr[b] + f) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int lr = 0; lr < 16; lr++) {
        const uint64_t row = rbase + lr;
        if (row >= M) break;
        float4 zc[NBA];
#pragma unroll
        for (int b = 0; b < NBA; b++) {
          zc[b] = zpre[b];
          if (lr + 1 < 16 && row + 1 < M && lane_on) zpre[b] = ld4s(d.Zr[b] + (row + 1) * d.ldzr[b] + f);
        }
        const float4 dy = lane_on ? *reinterpret_cast<const float4 *>(stash + lr * SP + f) : make_float4(0.f, 0.f, 0.f, 0.f);
        an_row_bwd<NBA>(d, row, f, lane_on, dy, zc, inv_seg, gs, go, gb);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (MODE == 1) {
    // per-workgroup partial sums of the parameter gradients: the four wavefronts' column sums are added in
    // wavefront order and left for act_norm_finish_kernel (fixed order: deterministic)
    float *red = reinterpret_cast<float *>(gsm);            // [4 waves][1 + 2 NBA][SP]
    constexpr int kKinds = 1 + 2 * NBA;
    __syncthreads();                                        // every wavefront is done with its stash
    if (lane_on) {
      float *rw = red + (size_t)wv * kKinds * SP + f;
      *reinterpret_cast<float4 *>(rw) = go;
#pragma unroll
      for (int b = 0; b < NBA; b++) {
        *reinterpret_cast<float4 *>(rw + (1 + 2 * b) * SP) = gs[b];
        *reinterpret_cast<float4 *>(rw + (2 + 2 * b) * SP) = gb[b];
      }
    }
    __syncthreads();
    for (uint32_t c = tid; c < d.N; c += kThreads) {
      float s[kKinds];
#pragma unroll
      for (int k = 0; k < kKinds; k++)
        s[k] = (red[(0 * kKinds + k) * SP + c] + red[(1 * kKinds + k) * SP + c]) + (red[(2 * kKinds + k) * SP + c] + red[(3 * kKinds + k) * SP + c]);
#pragma unroll
      for (int b = 0; b < NBA; b++) {
        float *pp = d.partial + (((size_t)blockIdx.x * NBA + b) * 3) * d.N + c;
        pp[0] = s[1 + 2 * b];                               // dscale
        pp[d.N] = s[0];                                     // doffset (the same sum for every branch)
        pp[2 * (size_t)d.N] = s[2 + 2 * b];                 // dbias
      }
    }
  }
}

int fill_dropout(FusedDesc &p, float drop_p, uint64_t drop_seed, const char *who) {
  p.drop_thr = 0; p.drop_scale = 1.0f; p.seed_lo = (uint32_t)drop_seed; p.seed_hi = (uint32_t)(drop_seed >> 32);
  if (drop_p <= 0.f) return SG_OK;
  if (!(drop_p < 1.f)) return set_error(SG_ERR_INVALID, "%s: dropout probability %g", who, drop_p);
  const double t = (double)drop_p * 4294967296.0;
  p.drop_thr = (uint32_t)std::min<double>(std::max<double>(t, 1.0), 4294967295.0);
  p.drop_scale = 1.0f / (1.0f - drop_p);
  return SG_OK;
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int TW, int MODE, int NBP, int NBA>
int launch_fused(const FusedDesc &d, hipStream_t st) {
  const size_t lds = (size_t)3 * 3 * TW * 64 * 16;
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

bool g_fused_epilogue = [] {
  const char *e = getenv("SHADOW_FUSED_EPILOGUE");
  return !(e && e[0] == '0');
}();

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

extern "C" int sl_gemm_act_norm_supported(uint32_t N, uint32_t K) {
  return g_fused_epilogue && N >= 16 && N <= 256 && (N & 3) == 0 && K > 0;
}

extern "C" int sl_gemm_act_norm_fwd(int nb, const float *const *d_A, const int64_t *lda, const void *d_packed_B, uint32_t M,
                                    uint32_t N, uint32_t K, float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                                    const int *act, const float *d_scale, const float *d_offset, float out_scale, float *d_out,
                                    int64_t ldo, float drop_p, uint64_t drop_seed, float *d_out_dropped, int64_t ldo_dropped,
                                    void *stream) {
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
    p.A[b] = d_A[b]; p.lda[b] = lda[b]; p.Z[b] = d_Z[b]; p.ldz[b] = ldz[b]; p.act[b] = act[b];
    p.bias[b] = d_bias ? d_bias[b] : nullptr;
  }
  if (!al16(d_scale) || !al16(d_offset) || !al16(d_out) || (ldo & 3))
    return set_error(SG_ERR_INVALID, "sl_gemm_act_norm_fwd: scale / offset / out must be 16-byte aligned, ldo %% 4 == 0");
  p.Bimg = reinterpret_cast<const bf16x8 *>(d_packed_B);
  p.M = M; p.N = N; p.K = K; p.units = (K + 31) / 32;
  p.scale = d_scale; p.offset = d_offset; p.out_scale = out_scale; p.eps = 1e-9f;
  p.out = d_out; p.ldo = ldo;
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

extern "C" size_t sl_gemm_an_bwd_partial_floats(uint32_t M, uint32_t N, int nb) {
  return (size_t)((M + 127) / 128) * (size_t)nb * 3 * N;
}

extern "C" int sl_gemm_an_bwd(const float *d_A, int64_t lda, const void *d_packed_B, uint32_t M, uint32_t N, uint32_t K, int nb,
                              const float *const *d_Z, const int64_t *ldz, const float *const *d_bias, const int *act,
                              const float *d_scale, const float *d_offset, float out_scale, float *const *d_dZ,
                              const int64_t *lddz, float *d_dscale, float *d_doffset, float *d_dbias, float *d_partial,
                              float drop_p, uint64_t drop_seed, void *stream) {
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
  p.A[0] = d_A; p.lda[0] = lda;
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
  p.Bimg = reinterpret_cast<const bf16x8 *>(d_packed_B);
  p.M = M; p.N = N; p.K = K; p.units = (K + 31) / 32;
  p.scale = d_scale; p.offset = d_offset; p.out_scale = out_scale; p.eps = 1e-9f;
  p.partial = d_partial;
  int rc;
  if ((rc = fill_dropout(p, drop_p, drop_seed, "sl_gemm_an_bwd")) != SG_OK) return rc;
  rc = N <= 128 ? launch_fused<4, 1, 1, 2>(p, st) : launch_fused<8, 1, 1, 2>(p, st);
  if (rc != SG_OK) return rc;
  return act_norm_finish_launch(d_partial, (M + 127) / 128, nb, N, d_dscale, d_doffset, d_dbias, st);
}
