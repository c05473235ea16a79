// gat.hip -- fused GAT attention for the shaDow GAT layer (shaDow/layers.py:560-626)
// behind the sl_gat_* C ABI.  All heads are processed together; a row of
// F = heads*D features is owned by LPR lanes (float4 each), a head slice by LS
// = D/4 lanes, so per-head reductions are LS-lane shuffles.
//
// forward  (per node)  hn = act(z_neigh), u_s = att[0]·act(z_self), u_n = att[1]·hn  per head
//          (per row)   e_ij = lrelu(u_s[i]) + lrelu(u_n[j])          layers.py:568-570
//                      softmax over the row with max subtraction,    layers.py:572-578
//                      numerator * drop-edge mask, denominator clamp 1e-10
//                      N_i = sum_j p_ij hn_j / den_i                 layers.py:580-581
// backward (per row)   t = dN_i·N_i per head (written by the act + norm backward that produces dN, or gat_t_kernel)
//          (per col)   alpha_ij recomputed from the row's (u_s, max, denominator); d e_ij = alpha_ij (dN_i·hn_j - t_i);
//                      d hn_j = sum_i alpha_ij dN_i + du_n att[1] ; dz_neigh ; datt[1]
//          du_s, the attention's share of dz_self and datt[0] are exactly zero (a row's weights do not change with u_s[i])
// (round 6: ONE edge walk instead of two -- the row walk that gathered hn_j per edge to leave alpha / de per edge behind
//  is gone: the column walk gathers dN_i anyway and holds hn_j in registers, so it forms dN_i·hn_j itself)
// The reference runs ~10 torch/scatter kernels per head per layer for this.
#include <string.h>

#include <algorithm>

#include "actnorm_common.h"
#include "common.h"
#include "gat_act.h"

namespace shadow {

constexpr int kGatBlock = 256;

__device__ __forceinline__ float4 gld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void gst4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float lrelu02(float x) { return x > 0.f ? x : 0.2f * x; }
__device__ __forceinline__ float dlrelu02(float x) { return x > 0.f ? 1.f : 0.2f; }

__device__ __forceinline__ float dot4(float4 a, float4 b) { return gat_dot4(a, b); }
// e^x of the edge softmax: the hardware exp2 on x log2(e) -- two instructions per edge and lane where libm's expf spends about ten
// on range handling these kernels do not need (arguments <= 0 up to rounding; a weight below 2^-126 is zero either way); relative
// error <= |x| 1e-7, forward and backward passes use the same function.  -DSHADOW_LIBM_EXP: expf.
__device__ __forceinline__ float gat_exp(float x) {
#ifdef SHADOW_LIBM_EXP
  return expf(x);
#else
  return __builtin_amdgcn_exp2f(x * 1.44269504f);
#endif
}

// Row ranges per XCD: workgroup b runs on XCD b % 8 (observed dispatch order, MI355X_MICROARCH.md; a wrong guess costs
// speed only).  The eight XCDs get contiguous eighths of the batch's rows and the workgroups of an XCD walk their eighth
// together, so that the feature rows a row's edges gather (same subgraph = neighbouring row ids, a few MB) are re-read
// out of that XCD's own 4 MB L2 instead of being spread over all eight.  (Grid-stride over the whole batch before: every
// XCD touched every subgraph.)  first / last / step for a block that handles `rpb` rows per iteration at offset `sub`.
struct RowWalk { uint64_t r, end, step; };
__device__ __forceinline__ RowWalk xcd_row_walk(uint32_t n, uint32_t rpb, uint32_t sub) {
  RowWalk w;
  if (gridDim.x % 8u != 0u) { w.r = (uint64_t)blockIdx.x * rpb + sub; w.end = n; w.step = (uint64_t)gridDim.x * rpb; return w; }
  const uint32_t xcd = blockIdx.x & 7u, bi = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const uint64_t per = (((uint64_t)n + 7) / 8 + rpb - 1) / rpb * rpb;       // rows per XCD, a multiple of rpb
  const uint64_t lo = (uint64_t)xcd * per;
  w.end = lo + per < n ? lo + per : n;
  w.r = lo + (uint64_t)bi * rpb + sub;
  w.step = (uint64_t)nbx * rpb;
  return w;
}

struct GatParams;
__device__ __forceinline__ float4 gat_hn_row(const GatParams &p, uint64_t row, uint32_t f);

struct GatParams {
  const uint32_t *indptr, *indices;       // CSR of the batch
  const uint32_t *t_indptr, *t_indices, *t_perm;
  const float *edge_w;                    // drop-edge mask or NULL
  const float *z_self, *z_neigh;          // [n, F] pre-activation
  const float *att;                       // [2, H, D]
  int act;
  uint32_t n, F, H, D;
  float *hn;                              // [n, F]  act(z_neigh), or NULL: recomputed from z_neigh wherever it is read (round 5)
  float *u_s, *u_n;                       // [n, H]  pre-leakyrelu scores
  float *mx, *den;                        // [n, H]
  float *nagg;                            // [n, F]  output
  // backward
  const float *dnagg;                     // [n, F]
  const float *tdot;                      // [n, H]  t_i = dN_i . N_i per head
  const uint32_t *dn_map;                 // sl_gat_bwd_map: dnagg / tdot are compact, row i's live at dn_map[i] (0xFFFFFFFF: a zero row)
  float *dz_self, *dz_neigh;              // [n, F]
  int acc_self;                           // dz_self already holds the act_norm branch's share of the gradient: add to it
  float *row_amax;                        // optional [n]: max |.| over the final dz_self and dz_neigh rows (sl_row_amax)
  float *datt;                            // [2, H, D]
  float *datt_part;                       // [gridDim.x][2][F] per-block sums, reduced in block order (gat_datt_finish_kernel)
  // forward tail (sl_gat_fwd_tail): out = out_scale * (norm_0(N) + norm_1(act(z_self))) per head slice, fused output dropout,
  // row maxima -- what sl_act_norm_fwd (nb = 2, seg = D) makes of (N, z_self), from the registers that hold N
  const float *scale, *offset;            // [2, F]: row 0 the aggregate's, row 1 the self branch's (shaDow/layers.py:620-622)
  float out_scale, eps, drop_scale;
  uint32_t drop_thr, seed_lo, seed_hi;
  float *out, *out_amax;                  // [n, F], optional [n]
};

// hn = act(z_neigh) of one row slice: the materialised copy, or -- hn not kept -- the activation applied where the row is
// gathered (one [n, F] write per forward pass and one read per backward pass less; the edge kernels wait for their gathers,
// the few extra VALU operations per gathered float4 ride in that shadow)
__device__ __forceinline__ float4 gat_hn_row(const GatParams &p, uint64_t row, uint32_t f) {
  return p.hn ? gld4(p.hn + row * p.F + f) : act4(p.act, gld4(p.z_neigh + row * p.F + f));
}

// Compile-time shape of a launch (round 6).  The edge walks are bound by instruction ISSUE (rocprofv3 SQ counters of the round-5
// kernels, profiles/r06_pmc_products-khop3-gat5_before.csv: the SIMDs issue 60 - 70 % of the kernel's cycles while every wavefront
// waits 80 - 90 % of its own), and with everything decided at run time the forward kernel's group-of-four loop body was 808 VALU
// instructions around 179 scalar branches (is there an edge mask?  is hn materialised?  which butterfly for this head width?).
// The configuration every GAT of config_train runs at the benchmark width -- 256 columns, 4 heads of 64, hn materialised -- is
// instantiated with those answers built in (edge mask: both ways); everything else takes the run-time form.
//   LS  lanes per head slice (0: p.D / 4 at run time)      W   1 / 0: edge mask present / absent, -1: look at p.edge_w
//   HN  1: hn is materialised, -1: look at p.hn             FIX 1: F == 256, H == 4 (D == 64): strides are shifts
template <int LS_, int W_, int HN_, int FIX_>
struct GatCfg { static constexpr int LS = LS_, W = W_, HN = HN_, FIX = FIX_; };
using GatDyn = GatCfg<0, -1, -1, 0>;

template <class C> __device__ __forceinline__ bool cfg_w(const GatParams &p) { if constexpr (C::W >= 0) return C::W != 0; else return p.edge_w != nullptr; }
template <class C> __device__ __forceinline__ uint32_t cfg_F(const GatParams &p) { if constexpr (C::FIX) return 256u; else return p.F; }
template <class C> __device__ __forceinline__ uint32_t cfg_H(const GatParams &p) { if constexpr (C::FIX) return 4u; else return p.H; }
template <class C> __device__ __forceinline__ float cfg_sum(float v, uint32_t ls) { if constexpr (C::LS > 0) return group_sum<C::LS>(v); else return slice_sum(v, ls); }
template <class C> __device__ __forceinline__ float4 cfg_hn_row(const GatParams &p, uint64_t row, uint32_t f) {
  if constexpr (C::HN == 1) return gld4(p.hn + row * cfg_F<C>(p) + f);
  else return gat_hn_row(p, row, f);
}

// The built-in shape gives a row to a WHOLE wavefront (256 columns = 64 lanes x float4): row ids, edge positions, column ids and
// mask values are the same in every lane.  hipcc cannot see that (they derive from threadIdx.x / 64), keeps them in vector registers
// and spends 64-bit vector arithmetic on every gather address: two thirds of the group-of-four loop body of the forward walk was
// address arithmetic.  uni*: the value from lane 0 (v_readfirstlane -> scalar registers, scalar address arithmetic, gathers in the
// scalar-base + lane-offset form); uld*: a load every lane would issue from the same address of data this launch only READS
// (structure, mask, per-node terms of an earlier launch) as ONE scalar load through the constant address space.
typedef const __attribute__((address_space(4))) uint32_t *gat_cu32p;
typedef const __attribute__((address_space(4))) float *gat_cf32p;
__device__ __forceinline__ uint32_t uni32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t uni64(uint64_t x) { return ((uint64_t)uni32((uint32_t)(x >> 32)) << 32) | uni32((uint32_t)x); }
template <class C> __device__ __forceinline__ uint32_t uld(const uint32_t *q, uint64_t i) {
  if constexpr (C::FIX) return ((gat_cu32p)(uintptr_t)q)[i]; else return q[i];
}
template <class C> __device__ __forceinline__ float uldf(const float *q, uint64_t i) {
  if constexpr (C::FIX) return ((gat_cf32p)(uintptr_t)q)[i]; else return q[i];
}
template <class C> __device__ __forceinline__ void uni_walk(RowWalk &w) {
  if constexpr (C::FIX) { w.r = uni64(w.r); w.end = uni64(w.end); w.step = uni64(w.step); }
}

template <int LPR>
__global__ void gat_node_fwd_kernel(GatParams p) {
  const uint32_t rpb = kGatBlock / LPR, sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t f = l * 4, ls = p.D / 4;
  const bool on = f < p.F;
  const uint32_t h = on ? f / p.D : 0;
  float4 a0 = make_float4(0, 0, 0, 0), a1 = a0;
  if (on) { a0 = gld4(p.att + f); a1 = gld4(p.att + p.F + f); }
  const RowWalk rw_ = xcd_row_walk(p.n, rpb, sub);
  for (uint64_t r = rw_.r; r < rw_.end; r += rw_.step) {
    float4 hs = make_float4(0, 0, 0, 0), hn = hs;
    if (on) {
      hs = act4(p.act, gld4(p.z_self + r * p.F + f));
      hn = act4(p.act, gld4(p.z_neigh + r * p.F + f));
      if (p.hn) gst4(p.hn + r * p.F + f, hn);
    }
    const float us = slice_sum(dot4(a0, hs), ls), un = slice_sum(dot4(a1, hn), ls);
    if (on && (l % ls) == 0) { p.u_s[r * p.H + h] = us; p.u_n[r * p.H + h] = un; }
  }
}


// Edge loops run in GROUPS: the G column ids, then the G scores, then the G feature rows of a group are loaded together, so a
// group costs one dependent round trip per stage instead of one per edge (sums keep the edge order: same bits as one by one).
// Online softmax (one pass instead of a maximum pass and a sum pass): the running maximum m is raised group by group and the sums
// so far are rescaled by exp(m_old - m_new).  mx / den come out as max_j e_j and sum_j exp(e_j - mx) w_j, to rounding.
template <int G, class C>
__device__ __forceinline__ void gat_fwd_group_online(const GatParams &p, uint32_t q, uint32_t h, uint32_t f, bool on, float as, float &m,
                                                     float &den, float4 &acc) {
  uint32_t c[G];
  float u[G], w[G];
  float4 v[G];
  const bool hw = cfg_w<C>(p);
#pragma unroll
  for (int j = 0; j < G; j++) { c[j] = p.indices[q + j]; w[j] = hw ? p.edge_w[q + j] : 1.0f; }
#pragma unroll
  for (int j = 0; j < G; j++) {
    u[j] = p.u_n[(uint64_t)c[j] * cfg_H<C>(p) + h];
    v[j] = on ? cfg_hn_row<C>(p, c[j], f) : make_float4(0, 0, 0, 0);
  }
  float gm = m;
#pragma unroll
  for (int j = 0; j < G; j++) { u[j] = as + lrelu02(u[j]); gm = fmaxf(gm, u[j]); }
  if (gm > m) {
    const float sc = gat_exp(m - gm);            // (m = -inf at the first group: exp(-inf) = 0, the sums are still zero)
    den *= sc; acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
    m = gm;
  }
#pragma unroll
  for (int j = 0; j < G; j++) {
    float pe = gat_exp(u[j] - m);
    if (hw) pe *= w[j];
    den += pe;
    acc.x += pe * v[j].x; acc.y += pe * v[j].y; acc.z += pe * v[j].z; acc.w += pe * v[j].w;
  }
}
// Column walk (transposed CSR): the edges (i -> j) into column j.  alpha_ij = exp(lrelu(u_s[i]) + lrelu(u_n[j]) - mx_i) w_ij / den_i
// is formed again from row i's three scalars -- the expression of the forward pass, same bits -- and de_ij = alpha_ij (dN_i.hn_j - t_i)
// with the gathered dN_i against the column's own hn_j (in registers).
// MAP (sl_gat_bwd_map): the incoming gradient lives on a few rows -- dnagg / tdot compact, row i's at dn_map[i] -- and an edge whose
// row has none (0xFFFFFFFF) is passed over: its terms are alpha_ij * 0 and 0, what the dense form adds for a zero row (with a whole
// wavefront per row -- 256 columns -- the test is wave-uniform and the five loads of such an edge are not issued).
template <int G, bool MAP, class C>
__device__ __forceinline__ void gat_bwd_col_group(const GatParams &p, uint32_t q, uint32_t h, uint32_t f, bool on, uint32_t ls, float lun,
                                                  const float4 &hn, float &dan, float4 &acc) {
  uint32_t s_[G], m_[G];
  float us[G], mx[G], dn[G], t[G], w[G];
  float4 v[G];
  const bool hw = cfg_w<C>(p);
#pragma unroll
  for (int j = 0; j < G; j++) { s_[j] = uld<C>(p.t_indices, q + j); w[j] = hw ? uldf<C>(p.edge_w, uld<C>(p.t_perm, q + j)) : 1.0f; }
#pragma unroll
  for (int j = 0; j < G; j++) m_[j] = MAP ? uld<C>(p.dn_map, s_[j]) : s_[j];
#pragma unroll
  for (int j = 0; j < G; j++) {
    us[j] = 0.f; mx[j] = 0.f; dn[j] = 1.f; t[j] = 0.f; v[j] = make_float4(0, 0, 0, 0);
    if (MAP && m_[j] == 0xFFFFFFFFu) continue;
    const uint64_t o = (uint64_t)s_[j] * cfg_H<C>(p) + h;
    us[j] = p.u_s[o]; mx[j] = p.mx[o]; dn[j] = p.den[o]; t[j] = p.tdot[(uint64_t)m_[j] * cfg_H<C>(p) + h];
    v[j] = on ? gld4(p.dnagg + (uint64_t)m_[j] * cfg_F<C>(p) + f) : make_float4(0, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < G; j++) {
    if (MAP && m_[j] == 0xFFFFFFFFu) continue;
    const float ev = lrelu02(us[j]) + lun;        // e_ij, the forward pass's expression: == mx_i exactly on row i's maximum edge
    float pe = gat_exp(ev - mx[j]);
    if (hw) pe *= w[j];
    const float alpha = pe * (1.0f / dn[j]);
    const float dal = cfg_sum<C>(dot4(v[j], hn), ls);
    // d e_ij.  Ordinary row: alpha_ij (dN_i . hn_j - t_i) -- the softmax's denominator moves with e_ij, the subtracted maximum
    // cancels.  A row whose denominator sits on its 1e-10 clamp: the denominator is a constant there, so the direct term is
    // alpha_ij dN_i . hn_j alone, and the subtracted row maximum no longer cancels: -t_i reaches the maximum edge (kept or dropped
    // by the edge mask) the way torch_scatter's max hands its gradient to the arg-max (shaDow/layers.py:572-578)
    const float tt = (dn[j] > 1e-10f) ? alpha * t[j] : (ev == mx[j] ? t[j] : 0.f);
    dan += alpha * dal - tt;
    acc.x += alpha * v[j].x; acc.y += alpha * v[j].y; acc.z += alpha * v[j].z; acc.w += alpha * v[j].w;
  }
}
// group sizes tried before the single-edge tail: bit masks of {8, 4, 2} per kernel (round 3, on the round-3 kernels; same box, products
// depth-3 GAT batches, gat_fwd / gat_bwd per launch and the step: one by one 641 / 1279 us, 13.73 ms; one mask for all three
// kernels {4} 505 / 1222, {4, 2} 481 / 1164, {2} 549 / 1101, {8, 4} 614 / 1241; forward {4, 2} with backward row / column
// {2} / {2} 1128 us, 12.52 ms; {4} / {2} 1106, 12.43; {2} / {4, 2} 1237; one by one / {2} 1233 -- the forward kernel takes the
// deeper groups, the backward kernels lose their occupancy to them)
#define SHADOW_GAT_GROUPS_FWD 6      // groups of 4, then 2, then single edges
#define SHADOW_GAT_GROUPS_COL 2      // groups of 2, then single edges
#define SHD_GAT_EDGES(MASK, q, b, CALL)                          \
  do {                                                           \
    if ((MASK) & 8) for (; q + 8 <= b; q += 8) { CALL(8); }      \
    if ((MASK) & 4) for (; q + 4 <= b; q += 4) { CALL(4); }      \
    if ((MASK) & 2) for (; q + 2 <= b; q += 2) { CALL(2); }      \
    for (; q < b; q++) { CALL(1); }                              \
  } while (0)

template <int LPR, bool TAIL, class C>
__global__ void gat_row_fwd_kernel(GatParams p) {
  const uint32_t rpb = kGatBlock / LPR, sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t f = l * 4, ls = C::LS > 0 ? (uint32_t)C::LS : p.D / 4;
  const uint32_t F = cfg_F<C>(p), H = cfg_H<C>(p);
  const bool on = f < F;
  const uint32_t h = on ? f / (C::FIX ? 64u : p.D) : 0;
  RowWalk rw_ = xcd_row_walk(p.n, rpb, sub);
  uni_walk<C>(rw_);
  // (the next row's pointers and score are loaded while this row's edges are walked: one dependent stage less per row)
  uint32_t na = 0, nb = 0;
  float nus = 0.f;
  if (rw_.r < rw_.end) { na = uld<C>(p.indptr, rw_.r); nb = uld<C>(p.indptr, rw_.r + 1); nus = p.u_s[rw_.r * H + h]; }
  for (uint64_t r = rw_.r; r < rw_.end; r += rw_.step) {
    const uint32_t a = na, b = nb;
    const float as = lrelu02(nus);
    if (r + rw_.step < rw_.end) { na = uld<C>(p.indptr, r + rw_.step); nb = uld<C>(p.indptr, r + rw_.step + 1); nus = p.u_s[(r + rw_.step) * H + h]; }
    // (tail: the row's own z_self slice is asked for AFTER the edge walk: before it, it is four more registers live through the walk --
    //  75 VGPRs, six wavefronts per SIMD, 437 us per launch against 360 with 71 / seven; docs/measurements/r06.md)
    float4 zs = make_float4(0, 0, 0, 0);
    float mx = -INFINITY;
    uint32_t q = a;
    float den = 0.f;
    float4 acc = make_float4(0, 0, 0, 0);
#define SHD_CALL(G) gat_fwd_group_online<G, C>(p, q, h, f, on, as, mx, den, acc)
    SHD_GAT_EDGES(SHADOW_GAT_GROUPS_FWD, q, b, SHD_CALL);
#undef SHD_CALL
    if (a == b) mx = 0.f;
    den = fmaxf(den, 1e-10f);
    const float inv = 1.0f / den;
    const float4 nv = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    if (on) {
      if (!TAIL || p.nagg) gst4(p.nagg + r * F + f, nv);
      if ((l % ls) == 0) { p.mx[r * H + h] = mx; p.den[r * H + h] = den; }
    }
    if (TAIL) {
      if (on) zs = ld4s(p.z_self + r * F + f);
      // out = out_scale * (norm_0(N) + norm_1(act(z_self))), each over the head slice (shaDow/layers.py:329-338,620-625): the
      // arithmetic of act_norm_kernel<.., false, 2> (aggregate.hip), statement for statement.  NOT bit for bit the separate pass:
      // hipcc contracts multiply-add pairs differently from one instantiation to the next (this kernel's edge walk uses packed
      // fmas where the plain row pass uses scalar ones, act_norm_kernel squares with v_pk_mul and adds unfused) -- the two forms
      // agree to 1 - 2 units in the last place per stage (tests: <= 2e-6 of the tensor's scale); each is reproducible run to run
      const float inv_seg = C::FIX ? 1.0f / 64.0f : 1.0f / (float)p.D;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t fo = f;
      asm volatile("" : "+v"(fo));       // (keeps the four loop-invariant scale / offset loads below from being hoisted over the edge walk: 16 VGPRs)
#pragma unroll
      for (int br = 0; br < 2; br++) {
        float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) hv = br == 0 ? nv : act4(p.act, zs);
        const float mean = cfg_sum<C>(hv.x + hv.y + hv.z + hv.w, ls) * inv_seg;
        float4 d = make_float4(hv.x - mean, hv.y - mean, hv.z - mean, hv.w - mean);
        if (!on) d = make_float4(0.f, 0.f, 0.f, 0.f);
        const float var = cfg_sum<C>(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w, ls) * inv_seg + p.eps;
        const float rstd = rsqrtf(var);
        float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), of = sc;
        if (on) { sc = gld4(p.scale + (size_t)br * F + fo); of = gld4(p.offset + (size_t)br * F + fo); }
        o.x += d.x * sc.x * rstd + of.x; o.y += d.y * sc.y * rstd + of.y;
        o.z += d.z * sc.z * rstd + of.z; o.w += d.w * sc.w * rstd + of.w;
      }
      float omax = 0.f;
      if (on) {
        o.x *= p.out_scale; o.y *= p.out_scale; o.z *= p.out_scale; o.w *= p.out_scale;
        omax = amax4(o);
        if (p.drop_thr) {
          const uint32_t keep = drop_keep4_raw(p.seed_lo, p.seed_hi, p.drop_thr, r, f);
          o = make_float4((keep & 1u) ? o.x * p.drop_scale : 0.f, (keep & 2u) ? o.y * p.drop_scale : 0.f,
                          (keep & 4u) ? o.z * p.drop_scale : 0.f, (keep & 8u) ? o.w * p.drop_scale : 0.f);
          omax = amax4(o);
        }
        st4s(p.out + r * F + f, o);
      }
      if (p.out_amax) {                   // (the LPR lanes of a row group share r)
        omax = group_max<LPR>(omax);
        if (l == 0) p.out_amax[r] = omax;
      }
    }
  }
}

// t_i = dN_i . N_i per head, for callers whose act + norm backward did not leave it (sl_gat_bwd with d_t == NULL)
template <int LPR>
__global__ void gat_t_kernel(GatParams p) {
  float *__restrict__ t_out = const_cast<float *>(p.tdot);
  const uint32_t rpb = kGatBlock / LPR, sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t f = l * 4, ls = p.D / 4;
  const bool on = f < p.F;
  const uint32_t h = on ? f / p.D : 0;
  for (uint64_t r = (uint64_t)blockIdx.x * rpb + sub; r < p.n; r += (uint64_t)gridDim.x * rpb) {
    float4 dn = make_float4(0, 0, 0, 0), ng = dn;
    if (on) { dn = ld4s(p.dnagg + r * p.F + f); ng = ld4s(p.nagg + r * p.F + f); }
    const float t = slice_sum(dot4(dn, ng), ls);
    if (on && (l % ls) == 0) t_out[r * p.H + h] = t;
  }
}

// backward, row side: there is none.  The score of edge (i, j) is lrelu(u_s[i]) + lrelu(u_n[j]) and the row's weights are
// exp(e_ij - max_j e_ij) over their (clamped) sum: adding a constant to every e_ij of a row -- which is all that moving u_s[i] does
// -- changes neither the numerators nor the sum, clamp or no clamp.  The aggregate's gradient with respect to u_s[i] is EXACTLY
// zero on every row; the reference's autograd arrives at rounding noise around zero (layers.py:568-581: sum_j de_ij = t_i - t_i
// for an ordinary row; sum_j alpha_ij dN_i.hn_j - t_i = 0 through the arg-max of torch_scatter's max for a row on the 1e-10 clamp).
// So the attention's share of dz_self and datt[0] are zero and z_self is not read at all by the backward pass (rounds 5 / 6 walked
// the clamped rows' edges row-wise and kept a sum there that the reference's graph does not have).
// backward, column side (transposed CSR): alpha, de, d hn, du_n, dz_neigh, datt[1]
// (round 6a: register caps -- amdgpu_waves_per_eu(6 / 7) -- measured and dropped: 578 us against 503 at the 87 VGPRs / 5 wavefronts hipcc
//  picked for the kernel of that time)
// (round 6b, same box: the row walk and the per-edge structure loads on the scalar side -- uni_walk / uld -- 446 -> 367 us per dense
//  launch, 70 -> 65 VGPRs; held to 64 for the eighth wavefront per SIMD: 373 -> 301 us, no spills.  The forward walk keeps its per-edge
//  loads in vector registers: with scalar ones it went from 359 to 397 us (one lgkmcnt(0) for all eight scalar loads of a group in
//  front of its gathers); its row walk alone on the scalar side: 353 -> 340 us, 71 -> 56 VGPRs.)
template <int LPR, bool MAP, class C>
__device__ __forceinline__ void gat_col_bwd_body(const GatParams &p) {
  const uint32_t rpb = kGatBlock / LPR, sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t f = l * 4, ls = C::LS > 0 ? (uint32_t)C::LS : p.D / 4;
  const uint32_t F = cfg_F<C>(p), H = cfg_H<C>(p);
  const bool on = f < F;
  const uint32_t h = on ? f / (C::FIX ? 64u : p.D) : 0;
  float4 a1 = make_float4(0, 0, 0, 0), g1 = a1;
  if (on) a1 = gld4(p.att + F + f);
  RowWalk rw_ = xcd_row_walk(p.n, rpb, sub);
  uni_walk<C>(rw_);
  uint32_t na = 0, nb = 0;                              // (next row's pointers ahead: see gat_row_fwd_kernel)
  if (rw_.r < rw_.end) { na = uld<C>(p.t_indptr, rw_.r); nb = uld<C>(p.t_indptr, rw_.r + 1); }
  for (uint64_t r = rw_.r; r < rw_.end; r += rw_.step) {
    const uint32_t a = na, b = nb;
    if (r + rw_.step < rw_.end) { na = uld<C>(p.t_indptr, r + rw_.step); nb = uld<C>(p.t_indptr, r + rw_.step + 1); }
    // the column's own hn_j and score (every edge's dN_i . hn_j and alpha_ij need them)
    float4 z = make_float4(0, 0, 0, 0), hn = z;
    if (on) {
      if (p.z_neigh) { z = gld4(p.z_neigh + r * F + f); hn = (C::HN == 1 || p.hn) ? gld4(p.hn + r * F + f) : act4(p.act, z); }
      else hn = gld4(p.hn + r * F + f);   // (the paired Linear wrote hn = act(z_neigh) straight away)
    }
    const float unr = p.u_n[r * H + h];
    const float lun = lrelu02(unr);
    float4 acc = make_float4(0, 0, 0, 0);
    float dan = 0.f, rmax = 0.f;
    uint32_t q = a;
#define SHD_CALL(G) gat_bwd_col_group<G, MAP, C>(p, q, h, f, on, ls, lun, hn, dan, acc)
    SHD_GAT_EDGES(SHADOW_GAT_GROUPS_COL, q, b, SHD_CALL);
#undef SHD_CALL
    if (on) {
      const float dun = dan * dlrelu02(unr);
      float4 dzv;
      acc.x += dun * a1.x; acc.y += dun * a1.y; acc.z += dun * a1.z; acc.w += dun * a1.w;
      if (p.z_neigh) {
        dzv = make_float4(acc.x * g_act_bwd(p.act, z.x, hn.x), acc.y * g_act_bwd(p.act, z.y, hn.y),
                          acc.z * g_act_bwd(p.act, z.z, hn.z), acc.w * g_act_bwd(p.act, z.w, hn.w));
      } else {                           // (the derivative from hn, gat_act.h)
        dzv = make_float4(acc.x * g_act_bwd_h(p.act, hn.x), acc.y * g_act_bwd_h(p.act, hn.y),
                          acc.z * g_act_bwd_h(p.act, hn.z), acc.w * g_act_bwd_h(p.act, hn.w));
      }
      gst4(p.dz_neigh + r * F + f, dzv);
      rmax = shadow::amax4(dzv);
      g1.x += dun * hn.x; g1.y += dun * hn.y; g1.z += dun * hn.z; g1.w += dun * hn.w;
    }
    if (p.row_amax) {                    // joined with the maximum of dz_self's row (accumulate mode: the caller's; otherwise zero rows)
      rmax = shadow::group_max<LPR>(rmax);
      if (l == 0) p.row_amax[r] = p.acc_self ? fmaxf(p.row_amax[r], rmax) : rmax;
    }
  }
  __shared__ float red[kGatBlock * 4];
  red[threadIdx.x * 4 + 0] = g1.x; red[threadIdx.x * 4 + 1] = g1.y; red[threadIdx.x * 4 + 2] = g1.z; red[threadIdx.x * 4 + 3] = g1.w;
  __syncthreads();
  if (sub == 0 && on) {
    float s4[4] = {0, 0, 0, 0};
    for (uint32_t q = 0; q < rpb; q++)
      for (int k = 0; k < 4; k++) s4[k] += red[(q * LPR + l) * 4 + k];
    for (int k = 0; k < 4; k++) {
      p.datt_part[((size_t)blockIdx.x * 2 + 1) * F + f + k] = s4[k];
      p.datt_part[((size_t)blockIdx.x * 2 + 0) * F + f + k] = 0.f;       // (datt[0]: no gradient through u_s, see above)
    }
  }
}

// the run-time shapes at the register count hipcc picks; the built-in shape held to 64 VGPRs (see above: the cap costs the
// run-time forms 5 - 7 spilled dwords, the built-in one none)
template <int LPR, bool MAP, class C>
__global__ void gat_col_bwd_kernel(GatParams p) { gat_col_bwd_body<LPR, MAP, C>(p); }
template <int LPR, bool MAP, class C>
__global__ void __attribute__((amdgpu_waves_per_eu(8))) gat_col_bwd_w8_kernel(GatParams p) { gat_col_bwd_body<LPR, MAP, C>(p); }

// (Round 3, measured and dropped: edge-parallel forms of the three edge kernels -- lane q owns edge q of a 64-edge chunk for
//  the scores / softmax, the numerators parked in LDS, the feature-row gathers of a chunk issued four at a time.  Slower
//  than the row-serial kernels above on the depth-3 products batches (forward 0.69 -> 0.72 ms, backward 1.06 -> 1.32 ms):
//  with e / n = 3.9 these kernels move 2.4 / 4.5 GB per launch counting the gathered rows, i.e. they already run at
//  3.5 - 4 TB/s of L2 / HBM traffic -- gather-volume-bound, not latency-bound.  What did help: the XCD-contiguous row walk.)
// (Round 5, measured and dropped: a head-split walk -- items (row, head) of D / 4 lanes in the order XCD range -> chunk of
//  4 096 rows -> head -> row, so that the slice re-touched between a row's first and last gather is 1 MB instead of the
//  subgraph's 4.6 MB of hn against a 4 MB L2.  Forward 0.47 -> 0.88 ms per launch at chunk heights 2 048 / 4 096 / 8 192
//  (same box, the products depth-3 GAT bench; the kernel was not kept: four times the wave iterations of a quarter of the work
//  each, four diverging items per wavefront): the row walk is bound by its dependent loads and per-edge arithmetic, not by where the
//  gathered rows come from -- recomputing elu where a row is gathered costs the same kernels 28 % (RECOMPUTE_HN).)
// datt[j] = sum over blocks of datt_part[block][j] in a fixed order (bit-reproducible; no float atomics).  A workgroup owns
// 32 outputs; its 32 x 32 threads cut the block range into 32 slices, eight independent running sums per thread (one thread
// per output walking all ~2 000 partial rows was a 240 us latency chain), the slices are added in slice order through LDS.
constexpr int kDattCols = 32, kDattSlices = 32;
__global__ void __launch_bounds__(kDattCols * kDattSlices)
gat_datt_finish_kernel(const float *__restrict__ part, uint32_t nblocks, uint32_t len, float *__restrict__ datt) {
  __shared__ float red[kDattSlices][kDattCols];
  const uint32_t c = threadIdx.x % kDattCols, sl = threadIdx.x / kDattCols;
  const uint32_t j = blockIdx.x * kDattCols + c;
  const uint32_t per = (nblocks + kDattSlices - 1) / kDattSlices;
  const uint32_t b0 = sl * per, b1 = min(nblocks, b0 + per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j < len) {
    uint32_t b = b0;
    for (; b + 8 <= b1; b += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) acc[k] += part[(size_t)(b + k) * len + j];
    }
    for (int k = 0; b < b1; b++, k++) acc[k] += part[(size_t)b * len + j];
  }
  red[sl][c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (sl == 0 && j < len) {
    float s = 0.f;
    for (int q = 0; q < kDattSlices; q++) s += red[q][c];
    datt[j] = s;
  }
}

static uint32_t gat_grid(uint32_t n, uint32_t lpr) {
  const uint32_t rpb = kGatBlock / lpr;
  uint64_t g = ((uint64_t)n + rpb - 1) / rpb;
  g = std::max<uint64_t>(1, std::min<uint64_t>(g, 256 * 8));
  if (g >= 8) g = (g + 7) & ~(uint64_t)7;                  // whole workgroups per XCD (xcd_row_walk)
  return (uint32_t)g;
}

static int gat_check(uint32_t F, uint32_t H, uint32_t *lpr_out) {
  if (H == 0 || F == 0 || F % H != 0) return set_error(SG_ERR_INVALID, "sl_gat: F=%u not divisible by heads=%u", F, H);
  const uint32_t D = F / H;
  if (D % 4 != 0 || ((D / 4) & (D / 4 - 1)) != 0 || F > 256)
    return set_error(SG_ERR_INVALID, "sl_gat: head slice D=%u must be 4*2^k and F=%u <= 256", D, F);
  uint32_t lpr = 4;
  while (lpr * 4 < F) lpr <<= 1;
  if (D / 4 > lpr) return set_error(SG_ERR_INVALID, "sl_gat: bad geometry");
  *lpr_out = lpr;
  return SG_OK;
}

#define SHD_GAT_LAUNCH(KERN, lpr, grid, st, p)                                            \
  do {                                                                                    \
    switch (lpr) {                                                                        \
      case 4: hipLaunchKernelGGL(KERN<4>, dim3(grid), dim3(kGatBlock), 0, st, p); break;  \
      case 8: hipLaunchKernelGGL(KERN<8>, dim3(grid), dim3(kGatBlock), 0, st, p); break;  \
      case 16: hipLaunchKernelGGL(KERN<16>, dim3(grid), dim3(kGatBlock), 0, st, p); break; \
      case 32: hipLaunchKernelGGL(KERN<32>, dim3(grid), dim3(kGatBlock), 0, st, p); break; \
      default: hipLaunchKernelGGL(KERN<64>, dim3(grid), dim3(kGatBlock), 0, st, p); break; \
    }                                                                                     \
  } while (0)

// the run-time form for every width, or -- 256 columns, 4 heads, hn materialised -- the built-in one (edge mask: both ways)
#define SHD_GAT_LAUNCH_CFG(KERN, lpr, grid, st, p, ...)                                                                   \
  do {                                                                                                                   \
    if ((p).F == 256 && (p).H == 4 && (p).hn) {                                                                          \
      if ((p).edge_w) hipLaunchKernelGGL((KERN<64, ##__VA_ARGS__, GatCfg<16, 1, 1, 1>>), dim3(grid), dim3(kGatBlock), 0, st, p); \
      else hipLaunchKernelGGL((KERN<64, ##__VA_ARGS__, GatCfg<16, 0, 1, 1>>), dim3(grid), dim3(kGatBlock), 0, st, p);     \
    } else {                                                                                                             \
      switch (lpr) {                                                                                                     \
        case 4: hipLaunchKernelGGL((KERN<4, ##__VA_ARGS__, GatDyn>), dim3(grid), dim3(kGatBlock), 0, st, p); break;       \
        case 8: hipLaunchKernelGGL((KERN<8, ##__VA_ARGS__, GatDyn>), dim3(grid), dim3(kGatBlock), 0, st, p); break;       \
        case 16: hipLaunchKernelGGL((KERN<16, ##__VA_ARGS__, GatDyn>), dim3(grid), dim3(kGatBlock), 0, st, p); break;     \
        case 32: hipLaunchKernelGGL((KERN<32, ##__VA_ARGS__, GatDyn>), dim3(grid), dim3(kGatBlock), 0, st, p); break;     \
        default: hipLaunchKernelGGL((KERN<64, ##__VA_ARGS__, GatDyn>), dim3(grid), dim3(kGatBlock), 0, st, p); break;     \
      }                                                                                                                  \
    }                                                                                                                    \
  } while (0)

}  // namespace shadow

using namespace shadow;

extern "C" int sl_gat_fwd(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                          const float *d_z_self, const float *d_z_neigh, const float *d_att, int act,
                          uint32_t n, uint32_t F, uint32_t heads, float *d_hn, float *d_u_s,
                          float *d_u_n, float *d_mx, float *d_den, float *d_nagg, void *stream_) {
  if (!d_indptr || !d_z_self || !d_z_neigh || !d_att || !d_u_s || !d_u_n || !d_mx || !d_den || !d_nagg)
    return set_error(SG_ERR_INVALID, "sl_gat_fwd: null argument");
  if (act < 0 || act > 4) return set_error(SG_ERR_INVALID, "sl_gat_fwd: unknown activation %d", act);
  uint32_t lpr;
  int rc = gat_check(F, heads, &lpr);
  if (rc) return rc;
  if (n == 0) return SG_OK;
  hipStream_t st = (hipStream_t)stream_;
  GatParams p;
  memset(&p, 0, sizeof(p));
  p.indptr = d_indptr; p.indices = d_indices; p.edge_w = d_edge_w; p.z_self = d_z_self; p.z_neigh = d_z_neigh;
  p.att = d_att; p.act = act; p.n = n; p.F = F; p.H = heads; p.D = F / heads;
  p.hn = d_hn; p.u_s = d_u_s; p.u_n = d_u_n; p.mx = d_mx; p.den = d_den; p.nagg = d_nagg;
  const uint32_t g = gat_grid(n, lpr);
  SHD_GAT_LAUNCH(gat_node_fwd_kernel, lpr, g, st, p);
  SHD_GAT_LAUNCH_CFG(gat_row_fwd_kernel, lpr, g, st, p, false);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

// The row pass alone: hn = act(z_neigh) and the per-node terms u_s / u_n are given (the paired Linear's GAT tail,
// sl_gemm_nt2_gat_f32, left them: gat_node_fwd_kernel's pass over z_self / z_neigh is not run).
extern "C" int sl_gat_fwd_rows(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w, const float *d_hn,
                               const float *d_u_s, const float *d_u_n, uint32_t n, uint32_t F, uint32_t heads, float *d_mx,
                               float *d_den, float *d_nagg, void *stream_) {
  if (!d_indptr || !d_hn || !d_u_s || !d_u_n || !d_mx || !d_den || !d_nagg) return set_error(SG_ERR_INVALID, "sl_gat_fwd_rows: null argument");
  uint32_t lpr;
  int rc = gat_check(F, heads, &lpr);
  if (rc) return rc;
  if (n == 0) return SG_OK;
  hipStream_t st = (hipStream_t)stream_;
  GatParams p;
  memset(&p, 0, sizeof(p));
  p.indptr = d_indptr; p.indices = d_indices; p.edge_w = d_edge_w; p.n = n; p.F = F; p.H = heads; p.D = F / heads;
  p.hn = const_cast<float *>(d_hn); p.u_s = const_cast<float *>(d_u_s); p.u_n = const_cast<float *>(d_u_n);
  p.mx = d_mx; p.den = d_den; p.nagg = d_nagg;
  const uint32_t g = gat_grid(n, lpr);
  SHD_GAT_LAUNCH_CFG(gat_row_fwd_kernel, lpr, g, st, p, false);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

// sl_gat_fwd_rows AND the layer's act + feature normalisation + branch average + output dropout in the same pass
// (shaDow/layers.py:612-625): the aggregate N of a row leaves the kernel normalised,
//   out = out_scale * (norm_0(N) + norm_1(act(z_self)))  per head slice, then the fused output dropout and the row maxima,
// what sl_act_norm_fwd (nb = 2, seg = F / heads, acts (identity, act)) makes of (N, z_self) -- same statements, equal to
// rounding -- without N going to memory and back in between.  d_nagg (optional) still receives N for the backward pass.
extern "C" int sl_gat_fwd_tail(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w, const float *d_hn,
                               const float *d_u_s, const float *d_u_n, const float *d_z_self, int act, const float *d_scale,
                               const float *d_offset, uint32_t n, uint32_t F, uint32_t heads, float out_scale, float drop_p,
                               uint64_t drop_seed, float *d_mx, float *d_den, float *d_nagg, float *d_out, float *d_out_amax,
                               void *stream_) {
  if (!d_indptr || !d_hn || !d_u_s || !d_u_n || !d_z_self || !d_scale || !d_offset || !d_mx || !d_den || !d_out)
    return set_error(SG_ERR_INVALID, "sl_gat_fwd_tail: null argument");
  if (act < 0 || act > 4) return set_error(SG_ERR_INVALID, "sl_gat_fwd_tail: unknown activation %d", act);
  uint32_t lpr;
  int rc = gat_check(F, heads, &lpr);
  if (rc) return rc;
  if (n == 0) return SG_OK;
  hipStream_t st = (hipStream_t)stream_;
  GatParams p;
  memset(&p, 0, sizeof(p));
  p.indptr = d_indptr; p.indices = d_indices; p.edge_w = d_edge_w; p.n = n; p.F = F; p.H = heads; p.D = F / heads;
  p.hn = const_cast<float *>(d_hn); p.u_s = const_cast<float *>(d_u_s); p.u_n = const_cast<float *>(d_u_n);
  p.z_self = d_z_self; p.act = act;
  p.mx = d_mx; p.den = d_den; p.nagg = d_nagg;
  p.scale = d_scale; p.offset = d_offset; p.out_scale = out_scale; p.eps = 1e-9f; p.out = d_out; p.out_amax = d_out_amax;
  p.drop_thr = 0; p.drop_scale = 1.0f; p.seed_lo = (uint32_t)drop_seed; p.seed_hi = (uint32_t)(drop_seed >> 32);
  if (drop_p > 0.f) {                    // (the threshold rule of sl_act_norm_fwd, aggregate.hip set_dropout)
    if (!(drop_p < 1.f)) return set_error(SG_ERR_INVALID, "sl_gat_fwd_tail: dropout probability %g", drop_p);
    p.drop_thr = drop_threshold16(drop_p);
    p.drop_scale = 1.0f / (1.0f - drop_p);
  }
  const uint32_t g = gat_grid(n, lpr);
  SHD_GAT_LAUNCH_CFG(gat_row_fwd_kernel, lpr, g, st, p, true);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

static int gat_bwd_impl(const uint32_t *d_indptr, const uint32_t *d_indices, const uint32_t *d_t_indptr,
                        const uint32_t *d_t_indices, const uint32_t *d_t_perm, const float *d_edge_w,
                        const float *d_z_self, const float *d_z_neigh, const float *d_att, int act,
                        uint32_t n, uint32_t e, uint32_t F, uint32_t heads, const float *d_hn,
                        const float *d_u_s, const float *d_u_n, const float *d_mx, const float *d_den,
                        const float *d_nagg, const float *d_dnagg, const float *d_t, const uint32_t *d_dn_map, float *d_work, float *d_dz_self,
                        float *d_dz_neigh, float *d_datt, int accumulate_dz_self, float *d_row_amax, void *stream_) {
  (void)e;
  if (!d_indptr || !d_t_indptr || !(d_z_neigh || d_hn) || !d_att || !d_u_s || !d_u_n || !d_mx ||
      !d_den || !d_nagg || !d_dnagg || !d_work || !d_dz_self || !d_dz_neigh || !d_datt)
    return set_error(SG_ERR_INVALID, "sl_gat_bwd: null argument");      // (d_z_neigh may be NULL when d_hn is given)
  uint32_t lpr;
  int rc = gat_check(F, heads, &lpr);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream_;
  if (n == 0) { SHD_HIP(hipMemsetAsync(d_datt, 0, (size_t)2 * F * 4, st)); return SG_OK; }
  GatParams p;
  memset(&p, 0, sizeof(p));
  p.indptr = d_indptr; p.indices = d_indices; p.t_indptr = d_t_indptr; p.t_indices = d_t_indices; p.t_perm = d_t_perm;
  p.edge_w = d_edge_w; p.z_self = d_z_self; p.z_neigh = d_z_neigh; p.att = d_att; p.act = act;
  p.n = n; p.F = F; p.H = heads; p.D = F / heads;
  p.hn = const_cast<float *>(d_hn); p.u_s = const_cast<float *>(d_u_s); p.u_n = const_cast<float *>(d_u_n);
  p.mx = const_cast<float *>(d_mx); p.den = const_cast<float *>(d_den); p.nagg = const_cast<float *>(d_nagg);
  p.dnagg = d_dnagg; p.dn_map = d_dn_map;
  // work: t[n*H] (when not given), datt_part[2048][2][F]
  p.datt_part = d_work + (size_t)n * heads;
  p.dz_self = d_dz_self; p.dz_neigh = d_dz_neigh; p.datt = d_datt; p.acc_self = accumulate_dz_self; p.row_amax = d_row_amax;
  const uint32_t g = gat_grid(n, lpr);
  if (d_t) p.tdot = d_t;
  else {
    p.tdot = d_work;
    SHD_GAT_LAUNCH(gat_t_kernel, lpr, g, st, p);
  }
  if (!accumulate_dz_self) SHD_HIP(hipMemsetAsync(d_dz_self, 0, (size_t)n * F * 4, st));     // (the attention's share of dz_self: zero)
  if (p.F == 256 && p.H == 4 && p.hn) {            // (the built-in shape of SHD_GAT_LAUNCH_CFG, on its eight-wavefront instantiation)
    if (d_dn_map) {
      if (p.edge_w) hipLaunchKernelGGL((gat_col_bwd_w8_kernel<64, true, GatCfg<16, 1, 1, 1>>), dim3(g), dim3(kGatBlock), 0, st, p);
      else hipLaunchKernelGGL((gat_col_bwd_w8_kernel<64, true, GatCfg<16, 0, 1, 1>>), dim3(g), dim3(kGatBlock), 0, st, p);
    } else {
      if (p.edge_w) hipLaunchKernelGGL((gat_col_bwd_w8_kernel<64, false, GatCfg<16, 1, 1, 1>>), dim3(g), dim3(kGatBlock), 0, st, p);
      else hipLaunchKernelGGL((gat_col_bwd_w8_kernel<64, false, GatCfg<16, 0, 1, 1>>), dim3(g), dim3(kGatBlock), 0, st, p);
    }
  } else if (d_dn_map) SHD_GAT_LAUNCH_CFG(gat_col_bwd_kernel, lpr, g, st, p, true);
  else SHD_GAT_LAUNCH_CFG(gat_col_bwd_kernel, lpr, g, st, p, false);
  hipLaunchKernelGGL(gat_datt_finish_kernel, dim3((2 * F + kDattCols - 1) / kDattCols), dim3(kDattCols * kDattSlices), 0, st,
                     p.datt_part, g, 2 * F, d_datt);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_gat_bwd(const uint32_t *d_indptr, const uint32_t *d_indices, const uint32_t *d_t_indptr,
                          const uint32_t *d_t_indices, const uint32_t *d_t_perm, const float *d_edge_w,
                          const float *d_z_self, const float *d_z_neigh, const float *d_att, int act,
                          uint32_t n, uint32_t e, uint32_t F, uint32_t heads, const float *d_hn,
                          const float *d_u_s, const float *d_u_n, const float *d_mx, const float *d_den,
                          const float *d_nagg, const float *d_dnagg, const float *d_t, float *d_work, float *d_dz_self,
                          float *d_dz_neigh, float *d_datt, int accumulate_dz_self, float *d_row_amax, void *stream_) {
  return gat_bwd_impl(d_indptr, d_indices, d_t_indptr, d_t_indices, d_t_perm, d_edge_w, d_z_self, d_z_neigh, d_att, act, n, e, F, heads, d_hn,
                      d_u_s, d_u_n, d_mx, d_den, d_nagg, d_dnagg, d_t, nullptr, d_work, d_dz_self, d_dz_neigh, d_datt, accumulate_dz_self,
                      d_row_amax, stream_);
}

extern "C" int sl_gat_bwd_map(const uint32_t *d_indptr, const uint32_t *d_indices, const uint32_t *d_t_indptr,
                              const uint32_t *d_t_indices, const uint32_t *d_t_perm, const float *d_edge_w,
                              const float *d_z_self, const float *d_z_neigh, const float *d_att, int act,
                              uint32_t n, uint32_t e, uint32_t F, uint32_t heads, const float *d_hn,
                              const float *d_u_s, const float *d_u_n, const float *d_mx, const float *d_den,
                              const float *d_nagg, const float *d_dnagg_rows, const float *d_t_rows, const uint32_t *d_dn_map,
                              float *d_work, float *d_dz_self, float *d_dz_neigh, float *d_datt, int accumulate_dz_self,
                              float *d_row_amax, void *stream_) {
  if (!d_dn_map || !d_t_rows) return set_error(SG_ERR_INVALID, "sl_gat_bwd_map: the row map and the rows' t are required");
  return gat_bwd_impl(d_indptr, d_indices, d_t_indptr, d_t_indices, d_t_perm, d_edge_w, d_z_self, d_z_neigh, d_att, act, n, e, F, heads, d_hn,
                      d_u_s, d_u_n, d_mx, d_den, d_nagg, d_dnagg_rows, d_t_rows, d_dn_map, d_work, d_dz_self, d_dz_neigh, d_datt,
                      accumulate_dz_self, d_row_amax, stream_);
}
