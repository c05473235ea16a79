// Dense feature x weight GEMMs of the layers (nn.Linear, shaDow/layers.py:433-435, :474-483, :604-611)
// on the bf16 matrix cores with fp32-level accuracy: the "split-bf16" scheme.
//
//   C[M,N] = A[M,K] . B[N,K]^T          A, B, C fp32 row-major (B = nn.Linear.weight, [out, in])
//
// Every fp32 operand is split EXACTLY into three bf16 pieces (8 + 8 + 8 mantissa bits):
//   x = x_h + x_m + x_l.
// The product keeps the six largest cross terms
//   a.b ~= a_h b_h + a_h b_m + a_m b_h + a_m b_m + a_h b_l + a_l b_h        (dropped: <= 2^-21 |a||b|)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The fp32 MFMA
// path runs at 1/16 of the bf16 rate on gfx950 (157 vs 2500 TFLOP/s), so six bf16 MFMAs are still
// ~2.7x faster than one fp32 MFMA pass, and for these skinny shapes (M ~ 3e5, N, K <= 256) the kernel
// ends up bound by streaming A in and C out of HBM (2 x 4 x M x 256 bytes) rather than by the matrix cores.
//
// Tiling (wave64, 32x32x16 MFMA):
//   workgroup = 4 wavefronts; a wavefront owns 32 rows x all N columns (N/32 MFMA tiles).
//   A is read straight from HBM into registers: lane (row r = lane & 31, half g = lane >> 5) loads the
//   16 contiguous floats A[row][32u + 16g .. +15] of "unit" u -- a wavefront load covers whole 128-byte
//   lines -- and splits them in registers.  The two MFMA k-steps of a unit therefore use the physical
//   columns  k = 32u + 16g + 8h + j  (h = step, j = 0..7); B is packed with the same permutation.
//   B is pre-split and pre-permuted once per call into the exact per-lane fragment image
//   [unit][step][piece][column tile][lane][8 x bf16] (sl_gemm_pack_b) and streamed through LDS one unit
//   (2 steps x 3 pieces x N/32 tiles x 1 KiB) at a time.
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "actnorm_common.h"
#include "common.h"
#include "gemm_common.h"

namespace shadow {
namespace {

#ifdef GEMM_TIMING
__device__ unsigned long long gemm_dbg[16];
#define GT_STAMP(i) do { if (blockIdx.x == 7 && threadIdx.x == 0) { const unsigned long long n_ = clock64(); gemm_dbg[i] += n_ - tl_; tl_ = n_; } } while (0)
#else
#define GT_STAMP(i) do {} while (0)
#endif

// ---------------------------------------------------------------------------
// B [N, K] fp32 -> fragment image: for unit u, step h, piece p, tile t, lane l: 8 bf16 =
// piece_p( B[32t + (l & 31)][32u + 16 (l >> 5) + 8h + j] ), j = 0..7.  Rows >= N and columns >= K are zero.
// One thread per (u, h, t, l).
// ---------------------------------------------------------------------------
// (B element (j, k) = B1[j * s1j + k * s1k] for k < K1, B2[j * s2j + (k - K1) * s2k] behind: a weight can be packed
//  transposed, and [Ws^T | Wn^T] without materialising the concatenation)
__global__ void gemm_pack_b_kernel(const float *__restrict__ B, int64_t s1j, int64_t s1k, uint32_t K1,
                                   const float *__restrict__ B2, int64_t s2j, int64_t s2k, uint32_t N, uint32_t K,
                                   uint32_t units, uint32_t tiles, bf16x8 *__restrict__ out) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = units * 2 * tiles * 64;
  if (idx >= total) return;
  const uint32_t l = idx & 63, t = (idx >> 6) % tiles, h = ((idx >> 6) / tiles) & 1, u = (idx >> 6) / tiles / 2;
  const uint32_t col = 32 * t + (l & 31);
  const uint32_t k0 = 32 * u + 16 * (l >> 5) + 8 * h;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t k = k0 + j;
    x[j] = (col < N && k < K) ? (k < K1 ? B[(int64_t)col * s1j + (int64_t)k * s1k] : B2[(int64_t)col * s2j + (int64_t)(k - K1) * s2k]) : 0.f;
  }
  bf16x8 hh, mm, ll;
  split8(x, hh, mm, ll);
  // image index: ((u*2 + h)*3 + p)*tiles + t, then lane
  const size_t base = ((size_t)(u * 2 + h) * 3) * tiles;
  out[(base + 0 * tiles + t) * 64 + l] = hh;
  out[(base + 1 * tiles + t) * 64 + l] = mm;
  out[(base + 2 * tiles + t) * 64 + l] = ll;
}

// ---------------------------------------------------------------------------
// fp16 two-piece images (gemm_common.h): the same fragment order with two pieces per step,
//   [unit][step][piece h, m][column tile][lane][8 x fp16],
// followed by a trailer of 2 x 32 tiles floats: the power-of-two scale of every output column's weight row (rows >= N: 1)
// and its inverse.  ONE launch packs all images of a layer pass (grid = tiles x images): a workgroup first finds the
// largest magnitude of its 32 weight rows (32 threads per row), then writes the tile's fragments of every k-step.
// `zero` (optional): n_zero floats cleared on the way (the row-maximum array the SpMM of the same pass joins into --
// saves the memset launch that the small batches, bound by the host's launch rate, would pay for).
struct PackSrc {
  const float *B1, *B2;          // element (j, k) = B1[j s1j + k s1k] for k < K1, B2[j s2j + (k - K1) s2k] behind
  int64_t s1j, s1k, s2j, s2k;
  uint32_t K1;
  half8 *img;
  float *trailer;
};
struct PackJob {
  PackSrc src[2];
  uint32_t N, K, units, tiles;
  float *zero;
  uint32_t n_zero;
};

__global__ void __launch_bounds__(1024) gemm_pack_f16_kernel(const PackJob job) {
  __shared__ float smax[32][32];
  __shared__ float sscale[32];
  const PackSrc &sr = job.src[blockIdx.y];
  const uint32_t t = blockIdx.x, tid = threadIdx.x, N = job.N, K = job.K, tiles = job.tiles;
  if (job.zero) {
    // (16 workgroups clear ~1.2 MB -- the aggregation's row-maximum array -- on a launch every layer pass waits for: 16-byte stores)
    const uint32_t nthr = gridDim.x * gridDim.y * 1024, me = (blockIdx.y * gridDim.x + blockIdx.x) * 1024 + tid;
    if ((reinterpret_cast<uintptr_t>(job.zero) & 15) == 0) {
      const uint32_t n4 = job.n_zero >> 2;
      float4 *z4 = reinterpret_cast<float4 *>(job.zero);
      for (uint32_t i = me; i < n4; i += nthr) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (uint32_t i = 4 * n4 + me; i < job.n_zero; i += nthr) job.zero[i] = 0.f;
    } else {
      for (uint32_t i = me; i < job.n_zero; i += nthr) job.zero[i] = 0.f;
    }
  }
  auto elem = [&](uint32_t col, uint32_t k) -> float {
    return k < sr.K1 ? sr.B1[(int64_t)col * sr.s1j + (int64_t)k * sr.s1k] : sr.B2[(int64_t)col * sr.s2j + (int64_t)(k - sr.K1) * sr.s2k];
  };
  {
    // 32 threads per weight row; the threads that sit next to each other run along the contiguous direction of the source
    // (a row-major weight: along k; a transposed one: along the rows)
    const bool kmajor = sr.s1k == 1;
    const uint32_t c = kmajor ? tid >> 5 : tid & 31u, kk = kmajor ? tid & 31u : tid >> 5, col = 32 * t + c;
    float mx = 0.f;
    if (col < N) {
      if (kmajor) {
#pragma unroll 4
        for (uint32_t k0 = 4 * kk; k0 < K; k0 += 128)
#pragma unroll
          for (uint32_t j = 0; j < 4; j++) if (k0 + j < K) mx = fmaxf(mx, fabsf(elem(col, k0 + j)));
      } else {
#pragma unroll 16
        for (uint32_t k = kk; k < K; k += 32) mx = fmaxf(mx, fabsf(elem(col, k)));
      }
    }
    smax[kk][c] = mx;
  }
  __syncthreads();
  if (tid < 32) {
    float mx = smax[0][tid];
#pragma unroll
    for (int q = 1; q < 32; q++) mx = fmaxf(mx, smax[q][tid]);
    const float sc = row_scale_of(mx);
    sscale[tid] = sc;
    sr.trailer[32 * t + tid] = sc;
    sr.trailer[32 * tiles + 32 * t + tid] = 1.0f / sc;    // (exact: a power of two within 2^+-62)
  }
  __syncthreads();
  // (all loads of a thread's fragments in flight before the first is used: the launch is a dependent step of every layer
  //  pass -- one round trip to the L2 instead of one per fragment)
  const uint32_t total = job.units * 2 * 64;
  for (uint32_t base = 0; base < total; base += 2 * 1024) {
    float x[2][8];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint32_t idx = base + q * 1024 + tid;
      const uint32_t l = idx & 63u, h = (idx >> 6) & 1u, u = idx >> 7;
      const uint32_t col = 32 * t + (l & 31u), k0 = 32 * u + 16 * (l >> 5) + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; j++) x[q][j] = (idx < total && col < N && k0 + j < K) ? elem(col, k0 + j) : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint32_t idx = base + q * 1024 + tid;
      if (idx >= total) continue;
      const uint32_t l = idx & 63u, h = (idx >> 6) & 1u, u = idx >> 7;
      half8 hh, mm;
      split8_f16(x[q], sscale[l & 31u], hh, mm);
      const size_t ob = ((size_t)(u * 2 + h) * 2) * tiles;
      sr.img[(ob + 0 * tiles + t) * 64 + l] = hh;
      sr.img[(ob + 1 * tiles + t) * 64 + l] = mm;
    }
  }
}

// Largest magnitude of every row of a row-major operand A [n, K] (lda % 4 == 0, 16-byte aligned): LPR lanes per row.
// The stand-alone form -- an extra pass over A; the kernels that PRODUCE an operand write the maxima while the row is in
// their registers (aggregate.hip, gemm_fused.hip) and this launch disappears.
template <int LPR>
__global__ void row_amax_kernel(const float *__restrict__ A, int64_t lda, uint32_t n, uint32_t K, float *__restrict__ amax) {
  constexpr int RP = 64 / LPR;
  const uint32_t lane = threadIdx.x & 63u, j = lane & (LPR - 1), rs = lane / LPR;
  const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  const uint32_t k4 = K / 4;
  for (uint64_t row0 = wave * (2 * RP); row0 < n; row0 += nwaves * (2 * RP)) {
    float mx[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const uint64_t row = row0 + i * RP + rs;
      if (row < n) {
        const float *ar = A + row * lda;
        for (uint32_t c = j; c < k4; c += LPR) mx[i] = fmaxf(mx[i], amax4(ld4(ar + 4 * c)));
        if (j == 0) for (uint32_t k = 4 * k4; k < K; k++) mx[i] = fmaxf(mx[i], fabsf(ar[k]));
      }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const float m = group_max<LPR>(mx[i]);
      const uint64_t row = row0 + i * RP + rs;
      if (j == 0 && row < n) amax[row] = m;
    }
  }
}

// ---------------------------------------------------------------------------
// Main kernel.  A wavefront owns RB x 32 rows and ALL NT column tiles (every A element is loaded by
// exactly one wavefront); a workgroup = 4 wavefronts.  The B image is streamed one k-step (16 columns)
// at a time, global -> LDS directly (global_load_lds, no VGPRs), into a ring of three buffers: the
// copies of step s+2 are issued during step s and only the OLDER ones are waited for (counted vmcnt +
// bare s_barrier), so no step ever waits for a copy it has just issued.  Measured on MI355X:
// the CU's vector-memory path (64 B/clk), not the matrix cores, is the scarce resource of this kernel, and
// a burst of loads right after a barrier stalls every wavefront on issue -- so the copies of the next
// step and the A registers of the next unit are issued one by one between the MFMA groups.  Two
// workgroups are resident per CU (register-bound): one keeps the matrix cores busy while the other sits
// in a barrier or in its epilogue.
// ---------------------------------------------------------------------------
// RB x 32 rows and TW column tiles per wavefront; the WAVES wavefronts of a workgroup form
// (WAVES / CS) row groups x CS column groups and share one B image ring (NT = TW * CS tiles).
// <1, 8, 1, 4>: 217 registers, 2 wavefronts per SIMD;  <1, 4, 2, 6>: half the accumulators per wavefront,
// 3 wavefronts per SIMD (A is then loaded by both column groups).
template <int RB, int TW, int CS, int WAVES, bool kTail>     // kTail: K % 32 != 0 (the last unit is zero-padded)
__global__ void __launch_bounds__(WAVES * 64, 2)
gemm_nt_split_kernel(const float *__restrict__ A, int64_t lda, const bf16x8 *__restrict__ Bimg, float *__restrict__ C,
                     int64_t ldc, uint32_t M, uint32_t N, uint32_t K, uint32_t units) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  constexpr int NT = TW * CS;                            // column tiles of the B image
  constexpr int kGemmThreads = WAVES * 64;
  constexpr int kRowsWG = 32 * RB * (WAVES / CS);
  constexpr int kPieces = 4 * RB;                        // 16-byte A pieces per lane per unit
  constexpr int kPPS = (kPieces + 2 * TW - 1) / (2 * TW);   // A pieces issued per (step, tile) slot
  constexpr int kStepVecs = 3 * NT * 64;                 // bf16x8 vectors of one k-step's B image
  constexpr int kFill = (kStepVecs + kGemmThreads - 1) / kGemmThreads;     // copies per thread per step
  constexpr int kFillPerSlot = (kFill + TW - 1) / TW;
  bf16x8 *lbuf = reinterpret_cast<bf16x8 *>(gsm);       // [3][kStepVecs]: ring of k-step images
  constexpr int kPieces0 = (kPieces < TW * kPPS) ? kPieces : TW * kPPS;   // A pieces issued during step h = 0 / h = 1
  constexpr int kPieces1 = kPieces - kPieces0;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t r = lane & 31u, g = lane >> 5;
  const uint32_t wrow = wv / CS, wcol = wv % CS;
  const uint64_t m0 = (uint64_t)blockIdx.x * kRowsWG + wrow * (32u * RB);
  const float *arow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; rb++) {
    uint64_t row = min(m0 + 32u * rb + r, (uint64_t)M - 1);   // rows past the end repeat the last row (never stored)
    arow[rb] = A + row * lda + 16 * g;
  }

  f32x16 acc[RB][TW];
#pragma unroll
  for (int rb = 0; rb < RB; rb++)
#pragma unroll
    for (int t = 0; t < TW; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[rb][t][i] = 0.f;

  float4 an[RB][4];                                      // A of the next unit
  // piece p = (row block p / 4, quarter p % 4) of unit u
  auto load_a_piece = [&](uint32_t u, int p) {
    const int rb = p >> 2, q = p & 3;
    const float *ptr = arow[rb] + 32 * u + 4 * q;
    if (!kTail || 32 * u + 32 <= K) {
      an[rb][q] = *reinterpret_cast<const float4 *>(ptr);
    } else {
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; c++) { const uint32_t k = 32 * u + 16 * g + 4 * q + c; v[c] = k < K ? ptr[c] : 0.f; }
      an[rb][q] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  // copies `slot` (of TW) of the B image of k-step s into its ring buffer
  auto fill_b = [&](uint32_t s, int slot) {
    const bf16x8 *src = Bimg + (size_t)s * kStepVecs;
    bf16x8 *dst = lbuf + (size_t)(s % 3u) * kStepVecs;
#pragma unroll
    for (int q = slot * kFillPerSlot; q < (slot + 1) * kFillPerSlot && q < kFill; q++) {
      const uint32_t base = q * kGemmThreads + wv * 64u;               // wave-uniform
      if (base < (uint32_t)kStepVecs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + base + lane),
                                         (__attribute__((address_space(3))) void *)(dst + base), 16, 0, 0);
    }
  };

#ifdef GEMM_TIMING
  unsigned long long tl_ = clock64();
  const unsigned long long t0_ = tl_, w0_ = wall_clock64();
#endif
#pragma unroll
  for (int slot = 0; slot < TW; slot++) fill_b(0, slot);
  if (units * 2 > 1) {
#pragma unroll
    for (int slot = 0; slot < TW; slot++) fill_b(1, slot);
  }
#pragma unroll
  for (int p = 0; p < kPieces; p++) load_a_piece(0, p);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  GT_STAMP(0);
  float4 ac[RB][4];
  const uint32_t steps = 2 * units;
  for (uint32_t u = 0; u < units; u++) {
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
      for (int q = 0; q < 4; q++) ac[rb][q] = an[rb][q];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t st = 2 * u + h;
      const bf16x8 *lb = lbuf + (size_t)(st % 3u) * kStepVecs;
      bf16x8 ah[RB], am[RB], al[RB];
#pragma unroll
      for (int rb = 0; rb < RB; rb++) {
        const float x[8] = {ac[rb][2 * h].x, ac[rb][2 * h].y, ac[rb][2 * h].z, ac[rb][2 * h].w,
                            ac[rb][2 * h + 1].x, ac[rb][2 * h + 1].y, ac[rb][2 * h + 1].z, ac[rb][2 * h + 1].w};
        split8(x, ah[rb], am[rb], al[rb]);
      }
      // (pin: the A registers are consumed -- and waited for -- BEFORE this step issues new copies; hipcc
      //  waits vmcnt(0) for an ordinary load while LDS-DMA copies are in flight)
      __builtin_amdgcn_sched_barrier(0);
      GT_STAMP(1);
      // B fragments are read one tile ahead of the MFMAs that consume them (by hand, with counted waits: see gemm_common.h)
      bf16x8 fb[2][3];
      const uint32_t la = lds_addr(lb + wcol * TW * 64 + lane);
      fb[0][0] = lds_read_frag<0>(la); fb[0][1] = lds_read_frag<NT * 1024>(la); fb[0][2] = lds_read_frag<2 * NT * 1024>(la);
#pragma unroll
      for (int t = 0; t < TW; t++) {
        if (t + 1 < TW) {
          const uint32_t lt = la + (t + 1) * 1024;          // (one VALU add per tile; the piece offsets are immediates)
          fb[(t + 1) & 1][0] = lds_read_frag<0>(lt);
          fb[(t + 1) & 1][1] = lds_read_frag<NT * 1024>(lt);
          fb[(t + 1) & 1][2] = lds_read_frag<2 * NT * 1024>(lt);
        }
        if (st + 2 < steps) fill_b(st + 2, t);              // two steps ahead: never waited for in this step
        if (u + 1 < units) {
#pragma unroll
          for (int p = (h * TW + t) * kPPS; p < (h * TW + t + 1) * kPPS && p < kPieces; p++) load_a_piece(u + 1, p);
        }
        if (t + 1 < TW) lds_wait<3>(); else lds_wait<0>();   // this tile's fragments have landed; the next tile's may still fly
        const bf16x8 bh = fb[t & 1][0], bm = fb[t & 1][1], bl = fb[t & 1][2];
        // small terms first, the dominant product last; row blocks alternate
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rb], bh, acc[rb][t], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rb], bl, acc[rb][t], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[rb], bm, acc[rb][t], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[rb], bh, acc[rb][t], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rb], bm, acc[rb][t], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rb], bh, acc[rb][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      GT_STAMP(2);
      // The next step's image was issued one step ago: wait only for what is OLDER than this step's own
      // copies (the vector-memory counter retires in order), then a bare barrier -- __syncthreads() would
      // drain the copies that are meant to stay in flight.
      if (u + 1 < units) {
        if (h == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kFill + kPieces0) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kFill + kPieces1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      GT_STAMP(3);
      __builtin_amdgcn_s_barrier();                      // every wavefront is done with this step's buffer
      GT_STAMP(4);
    }
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5).
  // Stored straight from that layout the tile leaves as 16 TW dword stores per lane, each covering two 128-byte row
  // pieces -- measured 17 % of a K = 256 workgroup's lifetime (scripts/probe_gemm_phases.sh), all workgroups at once.
  // The B ring is dead here: 16 rows of the tile at a time go through it (conflict-free ds_write_b32 in the C/D layout,
  // ds_read_b128 row-major) and leave as float4s, whole rows per instruction.
  const bool vec_ok = CS == 1 && (N & 3u) == 0 && (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
  if (vec_ok) {
    constexpr int SP = 32 * TW;                             // stash row pitch (floats)
    constexpr int F4 = 8 * TW;                              // float4s per row
    float *stash = reinterpret_cast<float *>(gsm) + (size_t)wv * (16 * SP);
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
#pragma unroll
        for (int t = 0; t < TW; t++)
#pragma unroll
          for (int ii = 0; ii < 8; ii++) stash[((ii & 3) + 8 * (ii >> 2) + 4 * g) * SP + 32 * t + r] = acc[rb][t][8 * hf + ii];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint64_t rbase = m0 + 32u * rb + 16u * hf;
#pragma unroll
        for (int it = 0; it < (16 * F4 + 63) / 64; it++) {
          const uint32_t idx = it * 64 + lane, lr = idx / F4, c4 = idx % F4;
          if (idx < 16 * F4 && rbase + lr < M && 4 * c4 < N)
            *reinterpret_cast<float4 *>(C + (rbase + lr) * ldc + 4 * c4) = *reinterpret_cast<const float4 *>(stash + lr * SP + 4 * c4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else {
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
#pragma unroll
      for (int t = 0; t < TW; t++) {
        const uint32_t col = 32 * (wcol * TW + t) + r;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const uint64_t rr = m0 + 32u * rb + (i & 3) + 8 * (i >> 2) + 4 * g;
          if (rr < M && col < N) C[rr * ldc + col] = acc[rb][t][i];
        }
      }
    }
  }
  GT_STAMP(5);
#ifdef GEMM_TIMING
  if (blockIdx.x == 7 && threadIdx.x == 0) { gemm_dbg[6] += clock64() - t0_; gemm_dbg[7] += wall_clock64() - w0_; gemm_dbg[8] += 1; }
#endif
}

}  // namespace
}  // namespace shadow

using namespace shadow;


extern "C" size_t sl_gemm_pack_bytes(uint32_t N, uint32_t K) {
  const size_t units = (K + 31) / 32, tiles = (N + 31) / 32;
  return units * 6 * tiles * 64 * 16;
}

extern "C" int sl_gemm_pack_b(const float *d_B, int64_t ldb, uint32_t N, uint32_t K, void *d_packed, void *stream) {
  if (!d_B || !d_packed) return set_error(SG_ERR_INVALID, "sl_gemm_pack_b: null argument");
  if (N == 0 || K == 0) return SG_OK;
  if (N > 256) return set_error(SG_ERR_INVALID, "sl_gemm_pack_b: N = %u (at most 256 output columns)", N);
  const uint32_t units = (K + 31) / 32, tiles = (N + 31) / 32;
  const uint32_t total = units * 2 * tiles * 64;
  hipLaunchKernelGGL(gemm_pack_b_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_B, ldb, (int64_t)1, K,
                     d_B, ldb, (int64_t)1, N, K, units, tiles, reinterpret_cast<bf16x8 *>(d_packed));
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_gemm_pack_b2(const float *d_B1, int64_t s1j, int64_t s1k, uint32_t K1, const float *d_B2, int64_t s2j,
                               int64_t s2k, uint32_t N, uint32_t K, void *d_packed, void *stream) {
  if (!d_B1 || !d_packed || (K1 < K && !d_B2)) return set_error(SG_ERR_INVALID, "sl_gemm_pack_b2: null argument");
  if (N == 0 || K == 0) return SG_OK;
  if (N > 256) return set_error(SG_ERR_INVALID, "sl_gemm_pack_b2: N = %u (at most 256 output columns)", N);
  const uint32_t units = (K + 31) / 32, tiles = (N + 31) / 32;
  const uint32_t total = units * 2 * tiles * 64;
  hipLaunchKernelGGL(gemm_pack_b_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_B1, s1j, s1k,
                     std::min(K1, K), d_B2 ? d_B2 : d_B1, s2j, s2k, N, K, units, tiles, reinterpret_cast<bf16x8 *>(d_packed));
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

// ---- fp16 two-piece operands (the kernels of gemm_fused.hip) ----
namespace shadow {
size_t pack_f16_image_bytes(uint32_t K, uint32_t tiles) { return (size_t)((K + 31) / 32) * 4 * tiles * 64 * 16; }
size_t pack_f16_trailer_bytes(uint32_t tiles) { return (size_t)2 * 32 * tiles * 4; }
// nimg <= 2 images of one launch: image b of B_b (strided / concatenated sources as sl_gemm_pack_b2) in `tiles` column tiles at
// d_img[b], its trailer at d_trailer[b]; zero / n_zero: floats cleared by the same launch (may be NULL)
int pack_f16(int nimg, const PackF16Src *src, uint32_t N, uint32_t K, uint32_t tiles, float *zero, uint32_t n_zero, hipStream_t st) {
  if (nimg < 1 || nimg > 2 || !src) return set_error(SG_ERR_INVALID, "fp16 weight pack: bad argument");
  if (N == 0 || K == 0) return SG_OK;
  if (N > 256 || tiles > 8 || 32 * tiles < N) return set_error(SG_ERR_INVALID, "fp16 weight pack: N = %u in %u column tiles", N, tiles);
  PackJob job;
  memset(&job, 0, sizeof(job));
  for (int b = 0; b < nimg; b++) {
    if (!src[b].B1 || !src[b].img || !src[b].trailer || (src[b].K1 < K && !src[b].B2))
      return set_error(SG_ERR_INVALID, "fp16 weight pack: null argument");
    PackSrc &d = job.src[b];
    d.B1 = src[b].B1; d.B2 = src[b].B2 ? src[b].B2 : src[b].B1; d.s1j = src[b].s1j; d.s1k = src[b].s1k; d.s2j = src[b].s2j; d.s2k = src[b].s2k;
    d.K1 = std::min(src[b].K1, K); d.img = reinterpret_cast<half8 *>(src[b].img); d.trailer = src[b].trailer;
  }
  job.N = N; job.K = K; job.units = (K + 31) / 32; job.tiles = tiles; job.zero = zero; job.n_zero = zero ? n_zero : 0;
  hipLaunchKernelGGL(gemm_pack_f16_kernel, dim3(tiles, nimg), dim3(1024), 0, st, job);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}
}  // namespace shadow

extern "C" int sl_row_amax(const float *d_A, int64_t lda, uint32_t n, uint32_t K, float *d_amax, void *stream) {
  if (!d_A || !d_amax) return set_error(SG_ERR_INVALID, "sl_row_amax: null argument");
  if (n == 0) return SG_OK;
  if ((lda & 3) || (reinterpret_cast<uintptr_t>(d_A) & 15)) return set_error(SG_ERR_INVALID, "sl_row_amax: A must be 16-byte aligned with lda %% 4 == 0");
  hipStream_t st = (hipStream_t)stream;
  SHD_PROF_FMT(4.0 * n * K + 4.0 * n, 0.0, st, "row_amax_K%u", K);
  const uint32_t k4 = K / 4;
  // (8 wavefront-rows in flight per wavefront and pass: 2 rows x 64 / LPR row slots)
  if (k4 > 32) {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 7) / 8, 256 * 16);
    hipLaunchKernelGGL(row_amax_kernel<64>, dim3(blocks), dim3(256), 0, st, d_A, lda, n, K, d_amax);
  } else if (k4 > 16) {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 15) / 16, 256 * 16);
    hipLaunchKernelGGL(row_amax_kernel<32>, dim3(blocks), dim3(256), 0, st, d_A, lda, n, K, d_amax);
  } else {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 31) / 32, 256 * 16);
    hipLaunchKernelGGL(row_amax_kernel<16>, dim3(blocks), dim3(256), 0, st, d_A, lda, n, K, d_amax);
  }
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_gemm_nt_f32(const float *d_A, int64_t lda, const void *d_packed_B, float *d_C, int64_t ldc,
                              uint32_t M, uint32_t N, uint32_t K, void *stream) {
  if (!d_A || !d_packed_B || !d_C) return set_error(SG_ERR_INVALID, "sl_gemm_nt_f32: null argument");
  if (M == 0 || N == 0) return SG_OK;
  if (K == 0) return set_error(SG_ERR_INVALID, "sl_gemm_nt_f32: K == 0");
  if (N > 256) return set_error(SG_ERR_INVALID, "sl_gemm_nt_f32: N = %u (at most 256 output columns)", N);
  if ((lda & 3) || (reinterpret_cast<uintptr_t>(d_A) & 15))
    return set_error(SG_ERR_INVALID, "sl_gemm_nt_f32: A must be 16-byte aligned with lda %% 4 == 0");
  const uint32_t units = (K + 31) / 32, tiles = (N + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
  const bf16x8 *img = reinterpret_cast<const bf16x8 *>(d_packed_B);
  const size_t lds = (size_t)3 * 3 * tiles * 64 * 16;
#define SHD_GEMM(RB, TW, CS, WAVES)                                                                          \
  {                                                                                                          \
    const uint32_t rows_wg = 32u * RB * (WAVES / CS);                                                        \
    const uint32_t grid = (M + rows_wg - 1) / rows_wg;                                                       \
    if (lds > 64 * 1024) {                                                                                   \
      SHD_HIP(ensure_dynamic_lds((const void *)gemm_nt_split_kernel<RB, TW, CS, WAVES, false>, lds));        \
      SHD_HIP(ensure_dynamic_lds((const void *)gemm_nt_split_kernel<RB, TW, CS, WAVES, true>, lds));         \
    }                                                                                                        \
    if (K % 32 == 0)                                                                                         \
      hipLaunchKernelGGL((gemm_nt_split_kernel<RB, TW, CS, WAVES, false>), dim3(grid), dim3(WAVES * 64), lds, st, d_A,  \
                         lda, img, d_C, ldc, M, N, K, units);                                        \
    else                                                                                                     \
      hipLaunchKernelGGL((gemm_nt_split_kernel<RB, TW, CS, WAVES, true>), dim3(grid), dim3(WAVES * 64), lds, st, d_A,   \
                         lda, img, d_C, ldc, M, N, K, units);                                        \
  }
  // (a column-split variant <1, 4, 2, 6> reaches 3 wavefronts per SIMD but measured slower: 0.35 vs 0.29 ms --
  //  A is loaded by both column groups and the vector-memory path is the scarce resource)
  switch (tiles) {
    case 1: SHD_GEMM(2, 1, 1, 4) break;
    case 2: SHD_GEMM(2, 2, 1, 4) break;
    case 3: SHD_GEMM(2, 3, 1, 4) break;
    case 4: SHD_GEMM(2, 4, 1, 4) break;
    case 5: SHD_GEMM(1, 5, 1, 4) break;
    case 6: SHD_GEMM(1, 6, 1, 4) break;
    case 7: SHD_GEMM(1, 7, 1, 4) break;
    default: SHD_GEMM(1, 8, 1, 4) break;
  }
#undef SHD_GEMM
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

#ifdef GEMM_TIMING
extern "C" int sl_gemm_debug_read(unsigned long long *out, int reset) {
  SHD_HIP(hipDeviceSynchronize());
  SHD_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(shadow::gemm_dbg), sizeof(unsigned long long) * 16));
  if (reset) { unsigned long long z[16] = {0}; SHD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(shadow::gemm_dbg), z, sizeof(z))); }
  return SG_OK;
}
#endif

// ===========================================================================
// Weight gradient  dW[N,K] = A[M,N]^T . B[M,K]   (A = dZ, B = layer input; M = batch nodes)
// ===========================================================================
// Same split-bf16 arithmetic; the reduction runs over the ROWS, so both operands are needed "k-major".
// Each workgroup owns a slice of the rows and walks it 16 rows (one MFMA k-step) at a time: the 16 x N
// and 16 x K fp32 row tiles go global -> LDS with global_load_lds (whole rows, lane-linear, double
// buffered, trickled between the MFMA groups); a fragment is eight ds_read_b32 down a column (one per
// row of the lane's half of the step), split into its three bf16 pieces in registers.  The 8 wavefronts
// form a 4 (N) x 2 (K) grid, each holding 2 x TK accumulator tiles.  Every workgroup writes its partial
// [N, K] product; gemm_tn_reduce_kernel adds the partials in a fixed order (deterministic).
namespace shadow {
namespace {

constexpr int kTnThreads = 512;
constexpr int kTnRowFloats = 256;                 // LDS row stride of both tiles (N, K <= 256)
constexpr int kTnStepFloats = 2 * 16 * kTnRowFloats;   // A tile + B tile of one k-step

__device__ __forceinline__ void lds_frag8(const float *tile, uint32_t col, uint32_t kg, float (&x)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) x[j] = tile[(8 * kg + j) * kTnRowFloats + col];
}

template <int TK>
__global__ void __launch_bounds__(kTnThreads)
gemm_tn_split_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
                     float *__restrict__ partial, uint32_t M, uint32_t N, uint32_t K, uint32_t rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  float *lbuf = reinterpret_cast<float *>(tsm);              // [2][kTnStepFloats]
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t r = lane & 31u, kg = lane >> 5;
  const uint32_t wn = wv >> 1, wk = wv & 1u;                  // 4 x 2 wavefront grid
  const uint64_t m_begin = (uint64_t)blockIdx.x * rows_per_wg;
  const uint64_t m_end = min((uint64_t)M, m_begin + rows_per_wg);
  const uint32_t steps = m_end > m_begin ? (uint32_t)((m_end - m_begin + 15) / 16) : 0u;   // (slices past M write zeros)

  // zero both buffers once: columns >= N / K and rows >= M are never written by the copies
  for (uint32_t i = tid; i < 2 * kTnStepFloats / 4; i += kTnThreads)
    reinterpret_cast<float4 *>(lbuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  f32x16 acc[2][TK];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int t = 0; t < TK; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[a][t][i] = 0.f;

  // copy `part` (0..3) of k-step s: this wavefront moves rows {2 wv, 2 wv + 1} of the A tile and of the B tile
  auto fill = [&](uint32_t s, int part) {
    const uint32_t lr = 2 * wv + (part & 1);                 // row inside the 16-row tile
    const uint64_t row = m_begin + (uint64_t)s * 16 + lr;
    const bool isb = part >= 2;
    float *dst = lbuf + (size_t)(s & 1) * kTnStepFloats + (isb ? 16 * kTnRowFloats : 0) + lr * kTnRowFloats;
    const uint32_t width = isb ? K : N;
    if (row < m_end) {
      const float *src = (isb ? B + row * ldb : A + row * lda);
      if (lane * 4 < width)     // (width % 4 == 0: whole 16-byte pieces)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane * 4),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    } else if (lane * 4 < width) {
      *reinterpret_cast<float4 *>(dst + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (steps > 0) {
#pragma unroll
    for (int part = 0; part < 4; part++) fill(0, part);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (uint32_t s = 0; s < steps; s++) {
    const float *at = lbuf + (size_t)(s & 1) * kTnStepFloats;
    const float *bt = at + 16 * kTnRowFloats;
    bf16x8 ah[2], am[2], al[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
      float x[8];
      lds_frag8(at, 32 * (2 * wn + a) + r, kg, x);
      split8(x, ah[a], am[a], al[a]);
    }
#pragma unroll
    for (int t = 0; t < TK; t++) {
      if (s + 1 < steps && t < 4) fill(s + 1, t);            // next step's rows trickle out between the MFMA groups
      if (TK < 4 && s + 1 < steps && t == TK - 1) {
#pragma unroll
        for (int part = TK; part < 4; part++) fill(s + 1, part);
      }
      float x[8];
      lds_frag8(bt, 32 * (TK * wk + t) + r, kg, x);
      bf16x8 bh, bm, bl;
      split8(x, bh, bm, bl);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bm, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bh, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bm, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // partial[g][n][k]; C/D layout: col (k) = lane & 31, row (n) = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
  float *out = partial + (size_t)blockIdx.x * N * K;
#pragma unroll
  for (int a = 0; a < 2; a++) {
#pragma unroll
    for (int t = 0; t < TK; t++) {
      const uint32_t kc = 32 * (TK * wk + t) + r;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint32_t nr = 32 * (2 * wn + a) + (i & 3) + 8 * (i >> 2) + 4 * kg;
        if (nr < N && kc < K) out[(size_t)nr * K + kc] = acc[a][t][i];
      }
    }
  }
}

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant expression
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The same product with every operand element split ONCE per workgroup.  In gemm_tn_split_kernel each wavefront splits
// the fragments it multiplies itself: 2 A tiles + TK B tiles per k-step -- 6 x ~60 VALU operations beside 48 MFMAs, and
// every A tile is split by two wavefronts, every B tile by four (3 x the necessary work; the kernel is as busy on the
// VALU as on the matrix cores).  Here wavefront w splits tile w of A and tile w of B of the NEXT k-step (8 LDS reads
// down a column, split8, three 16-byte LDS writes per fragment) into a double-buffered image of ready bf16 fragments
// [operand][piece][tile][lane], while everybody multiplies the current step out of the other image with plain
// ds_read_b128 fragment reads.  One barrier per step as before; LDS = 64 KB fp32 row tiles (two steps) + 96 KB fragment
// images (two steps) = all 160 KB of the CU (one 8-wave workgroup per CU, as before).  Same arithmetic in the same order:
// bit-identical products.
//
// kPipe (N = K = 256 only): the steps that have two full steps behind them run as ONE basic block -- no row / width
// predicates on the copies, no uniform branches -- whose instruction order is given by group barriers: one MFMA, then a few
// of the split's VALU operations and at most one LDS access, 48 times.  In the plain form the two wavefronts of a SIMD
// leave the step barrier together, split together (matrix pipe idle) and then multiply together (VALU idle): the parts of
// a step add up (3b of DESIGN.md).  Interleaved, the split runs in the shadow of the wavefront's own MFMAs.
template <int TK, bool kPipe>
__global__ void __launch_bounds__(kTnThreads)
gemm_tn_coop_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
                    float *__restrict__ partial, uint32_t M, uint32_t N, uint32_t K, uint32_t rows_per_wg, uint32_t colsum) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  float *lbuf = reinterpret_cast<float *>(tsm);                                   // [2][kTnStepFloats]: fp32 row tiles
  bf16x8 *fimg = reinterpret_cast<bf16x8 *>(tsm + (size_t)2 * kTnStepFloats * 4); // [2][2 operands][3 pieces][8 tiles][64 lanes]
  constexpr int kImgVecs = 2 * 3 * 8 * 64;                                        // bf16x8 vectors of one step's image
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t r = lane & 31u, kg = lane >> 5;
  const uint32_t wn = wv >> 1, wk = wv & 1u;                  // 4 x 2 wavefront grid
  const uint64_t m_begin = (uint64_t)blockIdx.x * rows_per_wg;
  const uint64_t m_end = min((uint64_t)M, m_begin + rows_per_wg);
  const uint32_t steps = m_end > m_begin ? (uint32_t)((m_end - m_begin + 15) / 16) : 0u;   // (slices past M write zeros)

  // zero both row-tile buffers once: columns >= N / K and rows >= M are never written by the copies
  for (uint32_t i = tid; i < 2 * kTnStepFloats / 4; i += kTnThreads)
    reinterpret_cast<float4 *>(lbuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  f32x16 acc[2][TK];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int t = 0; t < TK; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[a][t][i] = 0.f;

  float csum = 0.f;
  // copy `part` (0..3) of k-step s: this wavefront moves rows {2 wv, 2 wv + 1} of the A tile and of the B tile
  auto fill = [&](uint32_t s, int part) {
    const uint32_t lr = 2 * wv + (part & 1);                 // row inside the 16-row tile
    const uint64_t row = m_begin + (uint64_t)s * 16 + lr;
    const bool isb = part >= 2;
    float *dst = lbuf + (size_t)(s & 1) * kTnStepFloats + (isb ? 16 * kTnRowFloats : 0) + lr * kTnRowFloats;
    const uint32_t width = isb ? K : N;
    if (row < m_end) {
      const float *src = (isb ? B + row * ldb : A + row * lda);
      if (lane * 4 < width)     // (width % 4 == 0: whole 16-byte pieces)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane * 4),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    } else if (lane * 4 < width) {
      *reinterpret_cast<float4 *>(dst + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // fragment `lane` of tile `wv` of operand op (0: A, 1: B) of k-step s: row tile -> three bf16 pieces in the image
  auto split_job = [&](uint32_t s, int op) {
    const float *tile = lbuf + (size_t)(s & 1) * kTnStepFloats + (op ? 16 * kTnRowFloats : 0);
    float x[8];
    lds_frag8(tile, 32 * wv + r, kg, x);
    // (column sums of A -- nn.Linear's bias gradient -- ride along: this wavefront sees every element of its 32 columns once)
    if (colsum && op == 0) csum += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    bf16x8 h, m, l;
    split8(x, h, m, l);
    bf16x8 *dst = fimg + (size_t)(s & 1) * kImgVecs + ((size_t)op * 3 * 8 + wv) * 64 + lane;
    dst[0] = h; dst[8 * 64] = m; dst[2 * 8 * 64] = l;
  };
  if (steps > 0) {
#pragma unroll
    for (int part = 0; part < 4; part++) fill(0, part);
    if (steps > 1) {
#pragma unroll
      for (int part = 0; part < 4; part++) fill(1, part);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (steps > 0) { split_job(0, 0); split_job(0, 1); }
  __syncthreads();
  uint32_t s = 0;
  if constexpr (kPipe) {
    // steps s with s + 2 <= steps - 2: the rows of step s + 2 all lie below m_end, step s + 1 exists
    for (; s + 3 < steps; s++) {
      const bf16x8 *img = fimg + (size_t)(s & 1) * kImgVecs;
      {
        float *tb = lbuf + (size_t)(s & 1) * kTnStepFloats + (size_t)(2 * wv) * kTnRowFloats;
        const uint64_t row = m_begin + (uint64_t)(s + 2) * 16 + 2 * wv;
        const float *sa = A + row * lda + lane * 4, *sb = B + row * ldb + lane * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sa,
                                         (__attribute__((address_space(3))) void *)tb, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sa + lda),
                                         (__attribute__((address_space(3))) void *)(tb + kTnRowFloats), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sb,
                                         (__attribute__((address_space(3))) void *)(tb + 16 * kTnRowFloats), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + ldb),
                                         (__attribute__((address_space(3))) void *)(tb + 17 * kTnRowFloats), 16, 0, 0);
      }
      // Every LDS access of this block is issued by hand (invisible to hipcc's scoreboard, which would wait lgkmcnt(0)
      // before each first use) and waited for with counted waits -- LDS returns in order:
      //   top      F1 = al0 al1 bh | F2 = ah0 ah1 bl | F3 = am0 am1 bm  (tile 0, in the order the MFMA pairs need them),
      //            RA = the four raw reads of this wavefront's A column tile of step s + 1
      //   pair 0   wait F1 (10 younger accesses may stay in flight), pair 1: F2 (7), pair 2: F3 (4), then RA (0)
      //   pair 6t+3 issues the B fragments of tile t + 1 (waited for at pair 6t+6, nothing younger), pair 8 the raw reads of
      //   the B column tile; the pieces of the split follow the pairs two slots behind (piece p after pair p + 2).
      const uint32_t fa = lds_addr(img + (size_t)(2 * wn) * 64 + lane), fb = lds_addr(img + ((size_t)3 * 8 + TK * wk) * 64 + lane);
      const uint32_t ra = lds_addr(lbuf + (size_t)((s + 1) & 1) * kTnStepFloats + (size_t)(8 * kg) * kTnRowFloats + 32 * wv + r);
      const uint32_t wa = lds_addr(fimg + (size_t)((s + 1) & 1) * kImgVecs + (size_t)wv * 64 + lane);
      bf16x8 ah[2], am[2], al[2];
      bf16x8 bfr[2][3];                                           // B fragments of tile t (t & 1) and of the next one
      float2v xr[4], xq[4];                                       // raw columns of the A / B tile being split: rows 8 kg .. 8 kg + 7
      al[0] = lds_read_frag<16384>(fa); al[1] = lds_read_frag<16384 + 1024>(fa); bfr[0][0] = lds_read_frag<0>(fb);
      ah[0] = lds_read_frag<0>(fa); ah[1] = lds_read_frag<1024>(fa); bfr[0][2] = lds_read_frag<16384>(fb);
      am[0] = lds_read_frag<8192>(fa); am[1] = lds_read_frag<8192 + 1024>(fa); bfr[0][1] = lds_read_frag<8192>(fb);
      xr[0] = lds_read2st64<0, 4>(ra); xr[1] = lds_read2st64<8, 12>(ra); xr[2] = lds_read2st64<16, 20>(ra); xr[3] = lds_read2st64<24, 28>(ra);
      uint32_t hb[2], mb[2], lb[2];
      union { uint32_t u[4]; bf16x8 v; } H, Mm, L;
      auto piece = [&](auto pc) {
        constexpr int p = decltype(pc)::value, job = p / 12, q = p % 12, grp = q / 3, pos = q % 3;
        if constexpr (p == 0) {          // (column sums of A; unused without colsum)
          csum += ((xr[0].x + xr[0].y) + (xr[1].x + xr[1].y)) + ((xr[2].x + xr[2].y) + (xr[3].x + xr[3].y));
        }
        if constexpr (pos < 2) {
          const float x = job ? (pos ? xq[grp].y : xq[grp].x) : (pos ? xr[grp].y : xr[grp].x);
          hb[pos] = __float_as_uint(x) & 0xFFFF0000u;
          const float r1 = x - __uint_as_float(hb[pos]);
          mb[pos] = __float_as_uint(r1) & 0xFFFF0000u;
          lb[pos] = __float_as_uint(r1 - __uint_as_float(mb[pos]));
        } else {
          H.u[grp] = pack_hi16(hb[0], hb[1]);
          Mm.u[grp] = pack_hi16(mb[0], mb[1]);
          L.u[grp] = pack_hi16(lb[0], lb[1]);
        }
        if constexpr (q == 11) {
          lds_write_frag<job * 24576>(wa, H.v); lds_write_frag<job * 24576 + 8192>(wa, Mm.v); lds_write_frag<job * 24576 + 16384>(wa, L.v);
        }
      };
      __builtin_amdgcn_sched_barrier(0);
      static_for<6 * TK>([&](auto kc) {
        constexpr int k = decltype(kc)::value, t = k / 6, g = k % 6;
        if constexpr (k == 0) lds_wait<10>();
        else if constexpr (k == 1) lds_wait<7>();
        else if constexpr (k == 2) lds_wait<4>();
        else if constexpr (g == 0) lds_wait<0>();
        {
          const bf16x8 &bh = bfr[t & 1][0], &bm = bfr[t & 1][1], &bl = bfr[t & 1][2];
#pragma unroll
          for (int a = 0; a < 2; a++) {
            if constexpr (g == 0) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh, acc[a][t], 0, 0, 0);
            else if constexpr (g == 1) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl, acc[a][t], 0, 0, 0);
            else if constexpr (g == 2) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bm, acc[a][t], 0, 0, 0);
            else if constexpr (g == 3) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bh, acc[a][t], 0, 0, 0);
            else if constexpr (g == 4) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bm, acc[a][t], 0, 0, 0);
            else acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][t], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (k == 2) lds_wait<0>();                      // the raw A column
        if constexpr (g == 3 && t + 1 < TK) {
          constexpr int o = (t + 1) * 1024, nb = (t + 1) & 1;
          bfr[nb][0] = lds_read_frag<o>(fb); bfr[nb][1] = lds_read_frag<o + 8192>(fb); bfr[nb][2] = lds_read_frag<o + 16384>(fb);
        }
        if constexpr (k >= 2 && k < 22) piece(std::integral_constant<int, (k >= 2 && k < 22) ? k - 2 : 0>{});
        if constexpr (k == 22) { piece(std::integral_constant<int, 20>{}); piece(std::integral_constant<int, 21>{}); }
        if constexpr (k == 23) { piece(std::integral_constant<int, 22>{}); piece(std::integral_constant<int, 23>{}); }
        if constexpr (k == 8) {         // the raw B column: first used by piece 12 at pair 14, behind the wait of pair 12
          xq[0] = lds_read2st64<64, 68>(ra); xq[1] = lds_read2st64<72, 76>(ra); xq[2] = lds_read2st64<80, 84>(ra); xq[3] = lds_read2st64<88, 92>(ra);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  for (; s < steps; s++) {
    const bf16x8 *img = fimg + (size_t)(s & 1) * kImgVecs;
    // this step's A fragments (tiles 2 wn, 2 wn + 1)
    bf16x8 ah[2], am[2], al[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const bf16x8 *pa = img + (size_t)(2 * wn + a) * 64 + lane;
      ah[a] = pa[0]; am[a] = pa[8 * 64]; al[a] = pa[2 * 8 * 64];
    }
#pragma unroll
    for (int t = 0; t < TK; t++) {
      const bf16x8 *pb = img + ((size_t)3 * 8 + TK * wk + t) * 64 + lane;
      const bf16x8 bh = pb[0], bm = pb[8 * 64], bl = pb[2 * 8 * 64];
      // the row tiles of step s + 2 trickle in (their buffer was split during the previous step), and this wavefront's two
      // fragments of step s + 1 are split between the MFMA groups
      // the row tiles of step s + 2: all four copies at the top of the step -- they are waited for at its end, and a copy
      // issued between the later MFMA groups had less than half a step to arrive (233 -> 225 us at M = 289 k, N = K = 256)
      if (s + 2 < steps && t == 0) { fill(s + 2, 0); fill(s + 2, 1); fill(s + 2, 2); fill(s + 2, 3); }
      if (s + 1 < steps && t == 0) split_job(s + 1, 0);
      if (s + 1 < steps && t == TK / 2) split_job(s + 1, 1);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bm, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bh, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bm, acc[a][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // partial[g][n][k] (+ [N] column sums of A behind it when asked for); C/D layout: col (k) = lane & 31,
  // row (n) = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
  float *out = partial + (size_t)blockIdx.x * ((size_t)N * K + (colsum ? N : 0u));
  if (colsum) {
    const float both = csum + __shfl_xor(csum, 32, 64);            // the two 8-row halves of every step
    if (kg == 0 && 32 * wv + r < N) out[(size_t)N * K + 32 * wv + r] = both;
  }
#pragma unroll
  for (int a = 0; a < 2; a++) {
#pragma unroll
    for (int t = 0; t < TK; t++) {
      const uint32_t kc = 32 * (TK * wk + t) + r;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint32_t nr = 32 * (2 * wn + a) + (i & 3) + 8 * (i >> 2) + 4 * kg;
        if (nr < N && kc < K) out[(size_t)nr * K + kc] = acc[a][t][i];
      }
    }
  }
}

// out[i] = sum_g partial[g][i] in a fixed order: 8 independent running sums per thread (eight loads in
// flight instead of a dependent chain), 4 row-slice groups per output combined through LDS.
__global__ void __launch_bounds__(256)
gemm_tn_reduce_kernel(const float *__restrict__ partial, uint32_t G, uint32_t NK, float *__restrict__ out, uint32_t NK0,
                      float *__restrict__ out2,       // (outputs [NK0, NK) of a slice go to out2: the column sums)
                      float *__restrict__ outB = nullptr, float *__restrict__ out2B = nullptr) {   // blockIdx.y == 1: the pair launch's second product
  __shared__ float red[4][64];
  if (blockIdx.y) { partial += (size_t)G * NK; out = outB; out2 = out2B; }
  const uint32_t col = threadIdx.x & 63u, grp = threadIdx.x >> 6;       // 64 outputs x 4 slice groups per block
  const uint32_t i = blockIdx.x * 64u + col;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < NK) {
    const uint32_t per = (G + 3u) / 4u, g0 = grp * per, g1 = min(G, g0 + per);
    uint32_t g = g0;
    for (; g + 8 <= g1; g += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) acc[k] += partial[(size_t)(g + k) * NK + i];
    }
    for (int k = 0; g < g1; g++, k++) acc[k] += partial[(size_t)g * NK + i];
  }
  red[grp][col] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (grp == 0 && i < NK) {
    const float v = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    if (out2 && i >= NK0) out2[i - NK0] = v;
    else out[i] = v;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same product on TWO fp16 pieces per element and three MFMAs per tile (gemm_common.h: h = rn16(x), m = rn16(x - h);
// terms mh, hm, hh) -- half the matrix-core work the interleaved step above is bound by.  The contraction runs over the
// ROWS, so a row's scale cannot be taken out of the accumulators afterwards: row r of A is multiplied by s_r (the power of
// two that puts its largest magnitude into [2^14, 2^15)) and row r of B by t_r = 2^c / s_r, c one constant per row slice --
// the smallest s_r * (B's own top-of-range scale) over the slice's rows, so that no row of B leaves the fp16 range and the
// row pair with the largest product sits at the top of it; the partial product leaves the workgroup multiplied by 2^-c
// (all exact).  Row pairs more than 2^17 below the slice's largest lose low bits of B -- of terms that are below the sum's
// own fp32 resolution by then.  A row of zeros in either operand gets s_r = t_r = 0 (0 * inf never forms).  The caller
// supplies max_k |row| of both operands (any upper bound works; the producers' arrays of the nt kernels are used).
// N = K = 256; same slices, same partial layout and reduction as above; every step runs the hand-ordered block (the last
// steps with predicated copies): 12 slots of { MFMA pair | piece: two elements of the next step's split }.
typedef float float4v __attribute__((ext_vector_type(4)));
constexpr int kTnImg16Vecs = 2 * 2 * 8 * 64;          // half8 vectors of one step's image: [operand][piece][tile][lane]
constexpr uint32_t kTnF16MaxRows = 3040;              // rows of a slice whose scale pairs fit beside tiles and images

__device__ __forceinline__ int scale_exponent(float amax) {      // row_scale_of(amax) == 2^this (amax > 0)
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu) - 127;
  return min(max(14 - e, -62), 62);
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((uint32_t)(127 + e) << 23); }   // -126 <= e <= 127

__global__ void __launch_bounds__(kTnThreads)
gemm_tn_f16_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ aamax, const float *__restrict__ B, int64_t ldb,
                   const float *__restrict__ bamax, float *__restrict__ partial, uint32_t M, uint32_t rows_per_wg, uint32_t colsum,
                   const float *__restrict__ A2, uint32_t G) {
  constexpr int TK = 4;
  constexpr uint32_t N = 256, K = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  float *lbuf = reinterpret_cast<float *>(tsm);                                    // [2][kTnStepFloats]: fp32 row tiles
  half8 *fimg = reinterpret_cast<half8 *>(tsm + (size_t)2 * kTnStepFloats * 4);    // [2][kTnImg16Vecs]
  float *S = reinterpret_cast<float *>(tsm + (size_t)2 * kTnStepFloats * 4 + (size_t)2 * kTnImg16Vecs * 16);   // [rows_per_wg + 32]
  float *T = S + rows_per_wg + 32;
  __shared__ int red_e[8];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t r = lane & 31u, kg = lane >> 5;
  const uint32_t wn = wv >> 1, wk = wv & 1u;                  // 4 x 2 wavefront grid
  // Pair form (A2 != NULL, grid 2 G): TWO products against the same B -- A^T B and A2^T B, A2 with A's pitch and row maxima (the
  // two halves of the GraphSAGE backward's [dZs | A^T dZn] against X).  Workgroups b and b + 8 take the same row slice, one
  // product each: the dispatcher deals workgroups round-robin over the 8 XCDs, so the two sit on the same XCD, run in step
  // (same rows, same work) and the second one's B rows come out of that XCD's L2 instead of HBM.
  uint32_t slice = blockIdx.x, prod = 0;
  if (A2) { prod = (blockIdx.x >> 3) & 1u; slice = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7u); }
  if (prod) A = A2;
  const uint64_t m_begin = (uint64_t)slice * rows_per_wg;
  const uint64_t m_end = min((uint64_t)M, m_begin + rows_per_wg);
  const uint32_t steps = m_end > m_begin ? (uint32_t)((m_end - m_begin + 15) / 16) : 0u;   // (slices past M write zeros)

  for (uint32_t i = tid; i < 2 * kTnStepFloats / 4; i += kTnThreads)
    reinterpret_cast<float4 *>(lbuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // ---- the slice's scale pairs: exponents first (S holds e_A as an int, T the valid flag), then c, then the powers of two
  const uint32_t padded = (steps + 1) * 16;                   // (one step past the end: the last step splits a step nobody multiplies)
  int emin = 1 << 20;
  for (uint32_t i = tid; i < padded; i += kTnThreads) {
    const uint64_t row = m_begin + i;
    const float aa = row < m_end ? aamax[row] : 0.f, bb = row < m_end ? bamax[row] : 0.f;
    const bool valid = !(aa == 0.f) && !(bb == 0.f);          // (a NaN maximum counts: its row must reach the product, not vanish)
    const int ea = valid ? scale_exponent(aa) : 0, eb = valid ? scale_exponent(bb) : 0;
    if (valid) emin = min(emin, ea + eb);
    reinterpret_cast<int *>(S)[i] = ea;
    T[i] = valid ? 1.f : 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) emin = min(emin, __shfl_xor(emin, o, 64));
  if (lane == 0) red_e[wv] = emin;
  __syncthreads();
  int c = red_e[0];
#pragma unroll
  for (int w = 1; w < 8; w++) c = min(c, red_e[w]);
  if (c == (1 << 20)) c = 0;                                  // no row pair with a non-zero product
  for (uint32_t i = tid; i < padded; i += kTnThreads) {
    const int ea = reinterpret_cast<int *>(S)[i];
    const bool valid = T[i] != 0.f;
    const int te = c - ea;                                    // <= the row's own e_B: B stays inside the fp16 range
    S[i] = valid ? pow2i(ea) : 0.f;
    T[i] = (valid && te >= -126) ? pow2i(te) : 0.f;           // (below 2^-126: the pair is > 2^120 under the slice's largest)
  }

  f32x16 acc[2][TK];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int t = 0; t < TK; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[a][t][i] = 0.f;

  // copy `part` (0..3) of k-step s: this wavefront moves rows {2 wv, 2 wv + 1} of the A tile and of the B tile
  auto fill = [&](uint32_t s, int part) {
    const uint32_t lr = 2 * wv + (part & 1);
    const uint64_t row = m_begin + (uint64_t)s * 16 + lr;
    const bool isb = part >= 2;
    float *dst = lbuf + (size_t)(s & 1) * kTnStepFloats + (isb ? 16 * kTnRowFloats : 0) + lr * kTnRowFloats;
    if (row < m_end) {
      const float *src = (isb ? B + row * ldb : A + row * lda);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane * 4),
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    } else {
      *reinterpret_cast<float4 *>(dst + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float csum = 0.f;       // column sums of the RAW A values (nn.Linear's bias gradient): this wavefront sees every element of its 32 columns once
  // fragment `lane` of tile `wv` of operand op (0: A, 1: B) of k-step s: row tile -> two fp16 pieces in the image
  auto split_job16 = [&](uint32_t s, int op) {
    const float *tile = lbuf + (size_t)(s & 1) * kTnStepFloats + (op ? 16 * kTnRowFloats : 0);
    const float *sc = (op ? T : S) + (size_t)s * 16 + 8 * kg;
    half8 h, m;
    float raw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) raw[j] = tile[(8 * kg + j) * kTnRowFloats + 32 * wv + r];
    if (op == 0) csum += ((raw[0] + raw[1]) + (raw[2] + raw[3])) + ((raw[4] + raw[5]) + (raw[6] + raw[7]));
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float x = raw[j] * sc[j];
      const _Float16 hj = (_Float16)x;
      h[j] = hj;
      m[j] = (_Float16)(x - (float)hj);
    }
    half8 *dst = fimg + (size_t)(s & 1) * kTnImg16Vecs + ((size_t)op * 2 * 8 + wv) * 64 + lane;
    dst[0] = h; dst[8 * 64] = m;
  };
  __syncthreads();                                           // the zeroed tiles and the scale pairs are in place
  if (steps > 0) {
#pragma unroll
    for (int part = 0; part < 4; part++) fill(0, part);
    if (steps > 1) {
#pragma unroll
      for (int part = 0; part < 4; part++) fill(1, part);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (steps > 0) { split_job16(0, 0); split_job16(0, 1); }
  __syncthreads();

  half8 ah[2], am[2];
  half8 bfr[2][2];                                             // B fragments (h, m) of tile t (t & 1) and of the next one
  float2v xr[4], xq[4];                                        // raw columns of the A / B tile being split: rows 8 kg .. 8 kg + 7
  float4v sv[2], tv[2];                                        // their rows' scales
  auto body = [&](uint32_t s, auto steady_c) {
    constexpr bool steady = decltype(steady_c)::value;
    const half8 *img = fimg + (size_t)(s & 1) * kTnImg16Vecs;
    if constexpr (steady) {
      float *tb = lbuf + (size_t)(s & 1) * kTnStepFloats + (size_t)(2 * wv) * kTnRowFloats;
      const uint64_t row = m_begin + (uint64_t)(s + 2) * 16 + 2 * wv;
      const float *sa = A + row * lda + lane * 4, *sb = B + row * ldb + lane * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sa,
                                       (__attribute__((address_space(3))) void *)tb, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sa + lda),
                                       (__attribute__((address_space(3))) void *)(tb + kTnRowFloats), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sb,
                                       (__attribute__((address_space(3))) void *)(tb + 16 * kTnRowFloats), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + ldb),
                                       (__attribute__((address_space(3))) void *)(tb + 17 * kTnRowFloats), 16, 0, 0);
    } else if (s + 2 < steps) {
      fill(s + 2, 0); fill(s + 2, 1); fill(s + 2, 2); fill(s + 2, 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the zero fill of rows past the end is a plain LDS store)
    }
    // LDS accesses by hand, counted waits (see the bf16 kernel above).  Head: F1 = am0 am1 bh | F2 = ah0 ah1 bm (tile 0, in
    // the order the first two MFMA pairs need them), RA = raw A column of step s + 1, SA = the scales of its eight rows.
    const uint32_t fa = lds_addr(img + (size_t)(2 * wn) * 64 + lane), fb = lds_addr(img + ((size_t)2 * 8 + TK * wk) * 64 + lane);
    const uint32_t ra = lds_addr(lbuf + (size_t)((s + 1) & 1) * kTnStepFloats + (size_t)(8 * kg) * kTnRowFloats + 32 * wv + r);
    const uint32_t sa = lds_addr(S + (size_t)(s + 1) * 16 + 8 * kg), ta = lds_addr(T + (size_t)(s + 1) * 16 + 8 * kg);
    const uint32_t wa = lds_addr(fimg + (size_t)((s + 1) & 1) * kTnImg16Vecs + (size_t)wv * 64 + lane);
    am[0] = lds_read_frag<8192, half8>(fa); am[1] = lds_read_frag<8192 + 1024, half8>(fa); bfr[0][0] = lds_read_frag<0, half8>(fb);
    ah[0] = lds_read_frag<0, half8>(fa); ah[1] = lds_read_frag<1024, half8>(fa); bfr[0][1] = lds_read_frag<8192, half8>(fb);
    xr[0] = lds_read2st64<0, 4>(ra); xr[1] = lds_read2st64<8, 12>(ra); xr[2] = lds_read2st64<16, 20>(ra); xr[3] = lds_read2st64<24, 28>(ra);
    sv[0] = lds_read_frag<0, float4v>(sa); sv[1] = lds_read_frag<16, float4v>(sa);
    half8 H, Mm;
    auto piece = [&](auto pc) {                               // elements 2 q, 2 q + 1 of job (p / 4)
      constexpr int p = decltype(pc)::value, job = p / 4, q = p % 4;
      if constexpr (p == 0) {          // (the same order as split_job16; the step past the end holds stale data: not summed)
        const float s8 = ((xr[0].x + xr[0].y) + (xr[1].x + xr[1].y)) + ((xr[2].x + xr[2].y) + (xr[3].x + xr[3].y));
        csum += (s + 1 < steps) ? s8 : 0.f;
      }
      const float2v xx = job ? xq[q] : xr[q];
      const float4v ss = job ? tv[q / 2] : sv[q / 2];
      const float x0 = xx.x * ((q & 1) ? ss.z : ss.x), x1 = xx.y * ((q & 1) ? ss.w : ss.y);
      const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
      H[2 * q] = h0; H[2 * q + 1] = h1;
      Mm[2 * q] = (_Float16)(x0 - (float)h0); Mm[2 * q + 1] = (_Float16)(x1 - (float)h1);
      if constexpr (q == 3) { lds_write_frag<job * 16384>(wa, H); lds_write_frag<job * 16384 + 8192>(wa, Mm); }
    };
    __builtin_amdgcn_sched_barrier(0);
    static_for<3 * TK>([&](auto kc) {
      constexpr int k = decltype(kc)::value, t = k / 3, g = k % 3;
      if constexpr (k == 0) lds_wait<9>();
      else if constexpr (k == 1) lds_wait<6>();
      else if constexpr (g == 0) lds_wait<0>();
      {
        const half8 &bh = bfr[t & 1][0], &bm = bfr[t & 1][1];
#pragma unroll
        for (int a = 0; a < 2; a++) {
          if constexpr (g == 0) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[a], bh, acc[a][t], 0, 0, 0);
          else if constexpr (g == 1) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bm, acc[a][t], 0, 0, 0);
          else acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh, acc[a][t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (k == 1) lds_wait<0>();                      // the raw A column and its scales
      if constexpr (g == 1 && t + 1 < TK) {
        constexpr int o = (t + 1) * 1024, nb = (t + 1) & 1;
        bfr[nb][0] = lds_read_frag<o, half8>(fb); bfr[nb][1] = lds_read_frag<o + 8192, half8>(fb);
      }
      if constexpr (k == 2) {           // the raw B column and its scales: first used by piece 4 at pair 5, behind the wait of pair 3
        xq[0] = lds_read2st64<64, 68>(ra); xq[1] = lds_read2st64<72, 76>(ra); xq[2] = lds_read2st64<80, 84>(ra); xq[3] = lds_read2st64<88, 92>(ra);
        tv[0] = lds_read_frag<0, float4v>(ta); tv[1] = lds_read_frag<16, float4v>(ta);
      }
      if constexpr (k >= 1 && k <= 8) piece(std::integral_constant<int, (k >= 1 && k <= 8) ? k - 1 : 0>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  };
  uint32_t s = 0;
  for (; s + 3 < steps; s++) body(s, std::true_type{});
  for (; s < steps; s++) body(s, std::false_type{});

  // partial[g][n][k] = 2^-c * accumulators (+ [N] column sums of A behind it when asked for)
  const float unscale = pow2i(-c);
  float *out = partial + ((size_t)prod * G + slice) * ((size_t)N * K + (colsum ? N : 0u));
  if (colsum) {
    const float both = csum + __shfl_xor(csum, 32, 64);            // the two 8-row halves of every step
    if (kg == 0) out[(size_t)N * K + 32 * wv + r] = both;
  }
#pragma unroll
  for (int a = 0; a < 2; a++) {
#pragma unroll
    for (int t = 0; t < TK; t++) {
      const uint32_t kc = 32 * (TK * wk + t) + r;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint32_t nr = 32 * (2 * wn + a) + (i & 3) + 8 * (i >> 2) + 4 * kg;
        out[(size_t)nr * K + kc] = acc[a][t][i] * unscale;
      }
    }
  }
}

}  // namespace
}  // namespace shadow

// number of row slices (= partial products) sl_gemm_tn_f32 uses for M rows
extern "C" uint32_t sl_gemm_tn_slices(uint32_t M) {
  int ncu = 256, dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  // at least 8 k-steps per workgroup (256 until round 3: M = 39.5 k, N = K = 256 then ran on 154 CUs; 128 / 256 / 512 measured 76 / 80 /
  // 96 us per launch at 36 k rows)
  const uint32_t min_rows = 128u;
  return std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)ncu, (M + min_rows - 1) / min_rows));
}

// dW = A^T B on two fp16 pieces per element (gemm_tn_f16_kernel): N = K = 256, the row maxima of both operands given.
// Returns SG_ERR_INVALID for shapes the kernel does not take (callers fall back to sl_gemm_tn_f32).
extern "C" int sl_gemm_tn_f16(const float *d_A, int64_t lda, const float *d_a_amax, const float *d_B, int64_t ldb, const float *d_b_amax,
                              float *d_C, uint32_t M, uint32_t N, uint32_t K, float *d_partial, float *d_a_colsum, void *stream) {
  if (!d_A || !d_B || !d_C || !d_partial || !d_a_amax || !d_b_amax) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16: null argument");
  if (N != 256 || K != 256) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16: N = %u, K = %u (both 256)", N, K);
  if ((lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(d_A) & 15) || (reinterpret_cast<uintptr_t>(d_B) & 15))
    return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16: operands must be 16-byte aligned with ld %% 4 == 0");
  if (M == 0) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16: M = 0");
  hipStream_t st = (hipStream_t)stream;
  const uint32_t G = sl_gemm_tn_slices(M);
  uint32_t rows_per_wg = (M + G - 1) / G;
  rows_per_wg = (rows_per_wg + 15u) & ~15u;
  if (rows_per_wg > kTnF16MaxRows) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16: %u rows per slice (at most %u)", rows_per_wg, kTnF16MaxRows);
  const size_t lds = (size_t)2 * kTnStepFloats * 4 + (size_t)2 * kTnImg16Vecs * 16 + (size_t)2 * (rows_per_wg + 32) * 4;
  SHD_HIP(ensure_dynamic_lds((const void *)gemm_tn_f16_kernel, lds));
  hipLaunchKernelGGL(gemm_tn_f16_kernel, dim3(G), dim3(kTnThreads), lds, st, d_A, lda, d_a_amax, d_B, ldb, d_b_amax, d_partial, M, rows_per_wg,
                     d_a_colsum ? 1u : 0u, (const float *)nullptr, G);
  SHD_HIP(hipGetLastError());
  const uint32_t NK = N * K + (d_a_colsum ? N : 0u);
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((NK + 63) / 64), dim3(256), 0, st, d_partial, G, NK, d_C, N * K, d_a_colsum, (float *)nullptr, (float *)nullptr);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

// Two products against the same B in one launch: C1 = A1^T B, C2 = A2^T B (A2 with A1's pitch and row maxima); see the kernel.
// d_partial: 2 * sl_gemm_tn_slices(M) * (N * K [+ N with column sums]) floats.
extern "C" int sl_gemm_tn_f16_pair(const float *d_A1, const float *d_A2, int64_t lda, const float *d_a_amax, const float *d_B, int64_t ldb,
                                   const float *d_b_amax, float *d_C1, float *d_C2, uint32_t M, uint32_t N, uint32_t K, float *d_partial,
                                   float *d_a1_colsum, float *d_a2_colsum, void *stream) {
  if (!d_A1 || !d_A2 || !d_B || !d_C1 || !d_C2 || !d_partial || !d_a_amax || !d_b_amax) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16_pair: null argument");
  if (N != 256 || K != 256) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16_pair: N = %u, K = %u (both 256)", N, K);
  if ((lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(d_A1) & 15) || (reinterpret_cast<uintptr_t>(d_A2) & 15) || (reinterpret_cast<uintptr_t>(d_B) & 15))
    return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16_pair: operands must be 16-byte aligned with ld %% 4 == 0");
  if (M == 0) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16_pair: M = 0");
  hipStream_t st = (hipStream_t)stream;
  uint32_t G = sl_gemm_tn_slices(M);
  // The pair runs 2 G workgroups of ~137 KB LDS each -- one per CU: with G = #CUs that is two rounds, and every slice leaves a 256 KB
  // partial product per weight for the reduction to read back (2 x 64 MB written + read at 289 k rows).  Half the slices: one round of
  // twice the rows, half the partial traffic (SHADOW_GEMM_TN_PAIR_HALF=0: the full count; same sums per slice, the slices added in
  // slice order as before -- a different split of the rows, i.e. different rounding than G slices, deterministic either way).
  static const bool half_slices = [] { const char *e = getenv("SHADOW_GEMM_TN_PAIR_HALF"); return !(e && e[0] == '0'); }();
  if (half_slices && G >= 32 && (G / 2) % 8 == 0 && ((M + G / 2 - 1) / (G / 2) + 15u) / 16u * 16u <= kTnF16MaxRows) G /= 2;
  uint32_t rows_per_wg = (M + G - 1) / G;
  rows_per_wg = (rows_per_wg + 15u) & ~15u;
  if (rows_per_wg > kTnF16MaxRows) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16_pair: %u rows per slice (at most %u)", rows_per_wg, kTnF16MaxRows);
  if ((d_a1_colsum == nullptr) != (d_a2_colsum == nullptr)) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f16_pair: column sums for both products or for none");
  if (G % 8) {     // (the pairing of workgroups b and b + 8 needs whole groups of eight slices: small M -- two plain launches)
    int rc = sl_gemm_tn_f16(d_A1, lda, d_a_amax, d_B, ldb, d_b_amax, d_C1, M, N, K, d_partial, d_a1_colsum, stream);
    if (rc != SG_OK) return rc;
    return sl_gemm_tn_f16(d_A2, lda, d_a_amax, d_B, ldb, d_b_amax, d_C2, M, N, K, d_partial, d_a2_colsum, stream);
  }
  const size_t lds = (size_t)2 * kTnStepFloats * 4 + (size_t)2 * kTnImg16Vecs * 16 + (size_t)2 * (rows_per_wg + 32) * 4;
  SHD_HIP(ensure_dynamic_lds((const void *)gemm_tn_f16_kernel, lds));
  hipLaunchKernelGGL(gemm_tn_f16_kernel, dim3(2 * G), dim3(kTnThreads), lds, st, d_A1, lda, d_a_amax, d_B, ldb, d_b_amax, d_partial, M, rows_per_wg,
                     d_a1_colsum ? 1u : 0u, d_A2, G);
  SHD_HIP(hipGetLastError());
  const uint32_t NK = N * K + (d_a1_colsum ? N : 0u);
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((NK + 63) / 64, 2), dim3(256), 0, st, d_partial, G, NK, d_C1, N * K, d_a1_colsum, d_C2, d_a2_colsum);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_gemm_tn_f32(const float *d_A, int64_t lda, const float *d_B, int64_t ldb, float *d_C, uint32_t M,
                              uint32_t N, uint32_t K, float *d_partial, float *d_a_colsum, void *stream) {
  if (!d_A || !d_B || !d_C || !d_partial) return set_error(SG_ERR_INVALID, "sl_gemm_tn_f32: null argument");
  if (N == 0 || K == 0) return SG_OK;
  if (N > 256 || K > 256 || (N & 3) || (K & 3))
    return set_error(SG_ERR_INVALID, "sl_gemm_tn_f32: N = %u, K = %u (multiples of 4, at most 256)", N, K);
  if ((lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(d_A) & 15) || (reinterpret_cast<uintptr_t>(d_B) & 15))
    return set_error(SG_ERR_INVALID, "sl_gemm_tn_f32: operands must be 16-byte aligned with ld %% 4 == 0");
  hipStream_t st = (hipStream_t)stream;
  const uint32_t G = sl_gemm_tn_slices(M);
  uint32_t rows_per_wg = (M + G - 1) / G;
  rows_per_wg = (rows_per_wg + 15u) & ~15u;
  // (SHADOW_GEMM_TN_COOP=0: the per-wavefront-split kernel; read per call so that a test can compare the two in one process)
  // Measured (M = 289 k, N = 256, same box, scripts/probe_gemm_tn.py): K = 256: 241 -> 233 us; K = 128: 146 -> 156 us (the
  // fixed split work of the eight tile slots is not amortised by half the MFMAs) -- so the cooperative form runs for
  // K > 128 only.  Also measured and dropped: the row-tile copies three steps ahead in a four-buffer ring with counted
  // waits (235 us: HBM latency is not what holds the step either).
  const char *coop_env = getenv("SHADOW_GEMM_TN_COOP");
  const bool coop = d_a_colsum || (coop_env ? coop_env[0] != '0' : K > 128);     // (the column sums live in the cooperative kernel)
  if (coop) {
    const size_t ldsc = (size_t)2 * kTnStepFloats * 4 + (size_t)2 * 2 * 3 * 8 * 64 * 16;     // 64 KB row tiles + 96 KB fragment images
    if (K <= 128) {
      SHD_HIP(ensure_dynamic_lds((const void *)gemm_tn_coop_kernel<2, false>, ldsc));
      hipLaunchKernelGGL((gemm_tn_coop_kernel<2, false>), dim3(G), dim3(kTnThreads), ldsc, st, d_A, lda, d_B, ldb, d_partial, M, N, K, rows_per_wg, d_a_colsum ? 1u : 0u);
    } else {
      // (SHADOW_GEMM_TN_PIPE=0: the plain step loop; read per call so that a test can compare the two in one process)
      const char *pipe_env = getenv("SHADOW_GEMM_TN_PIPE");
      if (N == 256 && K == 256 && !(pipe_env && pipe_env[0] == '0')) {
        SHD_HIP(ensure_dynamic_lds((const void *)gemm_tn_coop_kernel<4, true>, ldsc));
        hipLaunchKernelGGL((gemm_tn_coop_kernel<4, true>), dim3(G), dim3(kTnThreads), ldsc, st, d_A, lda, d_B, ldb, d_partial, M, N, K, rows_per_wg, d_a_colsum ? 1u : 0u);
      } else {
        SHD_HIP(ensure_dynamic_lds((const void *)gemm_tn_coop_kernel<4, false>, ldsc));
        hipLaunchKernelGGL((gemm_tn_coop_kernel<4, false>), dim3(G), dim3(kTnThreads), ldsc, st, d_A, lda, d_B, ldb, d_partial, M, N, K, rows_per_wg, d_a_colsum ? 1u : 0u);
      }
    }
    SHD_HIP(hipGetLastError());
    const uint32_t NKc = N * K + (d_a_colsum ? N : 0u);
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((NKc + 63) / 64), dim3(256), 0, st, d_partial, G, NKc, d_C, N * K, d_a_colsum, (float *)nullptr, (float *)nullptr);
    SHD_HIP(hipGetLastError());
    return SG_OK;
  }
  const size_t lds = (size_t)2 * kTnStepFloats * 4;
  if (K <= 128) {
    SHD_HIP(ensure_dynamic_lds((const void *)gemm_tn_split_kernel<2>, lds));
    hipLaunchKernelGGL((gemm_tn_split_kernel<2>), dim3(G), dim3(kTnThreads), lds, st, d_A, lda, d_B, ldb, d_partial, M, N, K,
                       rows_per_wg);
  } else {
    SHD_HIP(ensure_dynamic_lds((const void *)gemm_tn_split_kernel<4>, lds));
    hipLaunchKernelGGL((gemm_tn_split_kernel<4>), dim3(G), dim3(kTnThreads), lds, st, d_A, lda, d_B, ldb, d_partial, M, N, K,
                       rows_per_wg);
  }
  SHD_HIP(hipGetLastError());
  const uint32_t NK = N * K;
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((NK + 63) / 64), dim3(256), 0, st, d_partial, G, NK, d_C, 0u, (float *)nullptr, (float *)nullptr, (float *)nullptr);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}
