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
// backward (per row)   t = dN_i·N_i ; d e_ij = alpha_ij (dN_i·hn_j - t) ; du_s ; dz_self (attention part)
//          (per col)   d hn_j = sum_i alpha_ij dN_i + du_n att[1] ; dz_neigh ; datt
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
  float *alpha, *de;                      // [e, H]
  float *du_s;                            // [n, H]
  float *dz_self, *dz_neigh;              // [n, F]
  int acc_self;                           // dz_self already holds the act_norm branch's share of the gradient: add to it
  float *row_amax;                        // optional [n]: max |.| over the final dz_self and dz_neigh rows (sl_row_amax)
  float *datt;                            // [2, H, D]
  float *datt_part;                       // [gridDim.x][2][F] per-block sums, reduced in block order (gat_datt_finish_kernel)
};

// hn = act(z_neigh) of one row slice: the materialised copy, or -- hn not kept -- the activation applied where the row is
// gathered (one [n, F] write per forward pass and one read per backward pass less; the edge kernels wait for their gathers,
// the few extra VALU operations per gathered float4 ride in that shadow)
__device__ __forceinline__ float4 gat_hn_row(const GatParams &p, uint64_t row, uint32_t f) {
  return p.hn ? gld4(p.hn + row * p.F + f) : act4(p.act, gld4(p.z_neigh + row * p.F + f));
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
template <int G>
__device__ __forceinline__ void gat_max_group(const GatParams &p, uint32_t q, uint32_t h, float as, float &mx) {
  uint32_t c[G];
  float u[G];
#pragma unroll
  for (int j = 0; j < G; j++) c[j] = p.indices[q + j];
#pragma unroll
  for (int j = 0; j < G; j++) u[j] = p.u_n[(uint64_t)c[j] * p.H + h];
#pragma unroll
  for (int j = 0; j < G; j++) mx = fmaxf(mx, as + lrelu02(u[j]));
}
template <int G>
__device__ __forceinline__ void gat_fwd_group(const GatParams &p, uint32_t q, uint32_t h, uint32_t f, bool on, float as, float mx,
                                              float &den, float4 &acc) {
  uint32_t c[G];
  float u[G], w[G];
  float4 v[G];
#pragma unroll
  for (int j = 0; j < G; j++) { c[j] = p.indices[q + j]; w[j] = p.edge_w ? p.edge_w[q + j] : 1.0f; }
#pragma unroll
  for (int j = 0; j < G; j++) {
    u[j] = p.u_n[(uint64_t)c[j] * p.H + h];
    v[j] = on ? gat_hn_row(p, c[j], f) : make_float4(0, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < G; j++) {
    float pe = expf(as + lrelu02(u[j]) - mx);
    if (p.edge_w) pe *= w[j];
    den += pe;
    acc.x += pe * v[j].x; acc.y += pe * v[j].y; acc.z += pe * v[j].z; acc.w += pe * v[j].w;
  }
}
// One pass instead of two (SHADOW_GAT_ONLINE_SOFTMAX): the running maximum m is raised group by group and the sums so far are
// rescaled by exp(m_old - m_new) -- the pass that only looked for the maximum (column ids -> scores: two dependent stages per
// group) goes away.  mx / den come out as max_j e_j and sum_j exp(e_j - mx) w_j like before, to rounding.
template <int G>
__device__ __forceinline__ void gat_fwd_group_online(const GatParams &p, uint32_t q, uint32_t h, uint32_t f, bool on, float as, float &m,
                                                     float &den, float4 &acc) {
  uint32_t c[G];
  float u[G], w[G];
  float4 v[G];
#pragma unroll
  for (int j = 0; j < G; j++) { c[j] = p.indices[q + j]; w[j] = p.edge_w ? p.edge_w[q + j] : 1.0f; }
#pragma unroll
  for (int j = 0; j < G; j++) {
    u[j] = p.u_n[(uint64_t)c[j] * p.H + h];
    v[j] = on ? gat_hn_row(p, c[j], f) : make_float4(0, 0, 0, 0);
  }
  float gm = m;
#pragma unroll
  for (int j = 0; j < G; j++) { u[j] = as + lrelu02(u[j]); gm = fmaxf(gm, u[j]); }
  if (gm > m) {
    const float sc = expf(m - gm);            // (m = -inf at the first group: exp(-inf) = 0, the sums are still zero)
    den *= sc; acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
    m = gm;
  }
#pragma unroll
  for (int j = 0; j < G; j++) {
    float pe = expf(u[j] - m);
    if (p.edge_w) pe *= w[j];
    den += pe;
    acc.x += pe * v[j].x; acc.y += pe * v[j].y; acc.z += pe * v[j].z; acc.w += pe * v[j].w;
  }
}
template <int G>
__device__ __forceinline__ void gat_bwd_row_group(const GatParams &p, uint32_t q, uint32_t h, uint32_t f, bool on, bool lead, uint32_t ls,
                                                  float as, float mx, float inv, float t, const float4 &dn, float &das) {
  uint32_t c[G];
  float u[G], w[G];
  float4 v[G];
#pragma unroll
  for (int j = 0; j < G; j++) { c[j] = p.indices[q + j]; w[j] = p.edge_w ? p.edge_w[q + j] : 1.0f; }
#pragma unroll
  for (int j = 0; j < G; j++) {
    u[j] = p.u_n[(uint64_t)c[j] * p.H + h];
    v[j] = on ? gat_hn_row(p, c[j], f) : make_float4(0, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < G; j++) {
    float pe = expf(as + lrelu02(u[j]) - mx);
    if (p.edge_w) pe *= w[j];
    const float alpha = pe * inv;
    const float dal = slice_sum(dot4(dn, v[j]), ls);
    const float de = alpha * (dal - t);
    das += de;
    if (lead) { p.alpha[(uint64_t)(q + j) * p.H + h] = alpha; p.de[(uint64_t)(q + j) * p.H + h] = de; }
  }
}
template <int G>
__device__ __forceinline__ void gat_bwd_col_group(const GatParams &p, uint32_t q, uint32_t h, uint32_t f, bool on, float &dan, float4 &acc) {
  uint32_t s_[G], e_[G];
  float al[G], de[G];
  float4 v[G];
#pragma unroll
  for (int j = 0; j < G; j++) { s_[j] = p.t_indices[q + j]; e_[j] = p.t_perm[q + j]; }
#pragma unroll
  for (int j = 0; j < G; j++) {
    al[j] = p.alpha[(uint64_t)e_[j] * p.H + h];
    de[j] = p.de[(uint64_t)e_[j] * p.H + h];
    v[j] = on ? gld4(p.dnagg + (uint64_t)s_[j] * p.F + f) : make_float4(0, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < G; j++) {
    dan += de[j];
    acc.x += al[j] * v[j].x; acc.y += al[j] * v[j].y; acc.z += al[j] * v[j].z; acc.w += al[j] * v[j].w;
  }
}
// group sizes tried before the single-edge tail: bit masks of {8, 4, 2} per kernel (scripts/ab_gat_group.sh; same box, products
// depth-3 GAT batches, gat_fwd / gat_bwd per launch and the step: one by one 641 / 1279 us, 13.73 ms; one mask for all three
// kernels {4} 505 / 1222, {4, 2} 481 / 1164, {2} 549 / 1101, {8, 4} 614 / 1241; forward {4, 2} with backward row / column
// {2} / {2} 1128 us, 12.52 ms; {4} / {2} 1106, 12.43; {2} / {4, 2} 1237; one by one / {2} 1233 -- the forward kernel takes the
// deeper groups, the backward kernels lose their occupancy to them)
#ifndef SHADOW_GAT_GROUPS_FWD
#define SHADOW_GAT_GROUPS_FWD 6
#endif
#ifndef SHADOW_GAT_GROUPS_ROW
#define SHADOW_GAT_GROUPS_ROW 2      // (round 5, after the row pass stopped reading z_self: groups of two -- 92 VGPRs instead of 106 -- 0.87 -> 0.82 ms per gat_bwd launch, step 10.42 -> 10.24 ms, same box twice: scripts/micro/ab_gat_row_waves.sh; {4} was round 3's choice)
#endif
#ifndef SHADOW_GAT_GROUPS_COL
#define SHADOW_GAT_GROUPS_COL 2
#endif
#ifndef SHADOW_GAT_ONLINE_SOFTMAX
#define SHADOW_GAT_ONLINE_SOFTMAX 1
#endif
#ifndef SHADOW_GAT_ROW_PREFETCH
#define SHADOW_GAT_ROW_PREFETCH 1      // the next row's pointers / score / gradient rows are loaded while this row's edges are walked
#endif
#define SHD_GAT_EDGES(MASK, q, b, CALL)                          \
  do {                                                           \
    if ((MASK) & 8) for (; q + 8 <= b; q += 8) { CALL(8); }      \
    if ((MASK) & 4) for (; q + 4 <= b; q += 4) { CALL(4); }      \
    if ((MASK) & 2) for (; q + 2 <= b; q += 2) { CALL(2); }      \
    for (; q < b; q++) { CALL(1); }                              \
  } while (0)

template <int LPR>
__global__ void gat_row_fwd_kernel(GatParams p) {
  const uint32_t rpb = kGatBlock / LPR, sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t f = l * 4, ls = p.D / 4;
  const bool on = f < p.F;
  const uint32_t h = on ? f / p.D : 0;
  const RowWalk rw_ = xcd_row_walk(p.n, rpb, sub);
  // (the next row's pointers and score are loaded while this row's edges are walked: one dependent stage less per row)
  uint32_t na = 0, nb = 0;
  float nus = 0.f;
  if (rw_.r < rw_.end) { na = p.indptr[rw_.r]; nb = p.indptr[rw_.r + 1]; nus = p.u_s[rw_.r * p.H + h]; }
  for (uint64_t r = rw_.r; r < rw_.end; r += rw_.step) {
#if !SHADOW_GAT_ROW_PREFETCH
    na = p.indptr[r]; nb = p.indptr[r + 1]; nus = p.u_s[r * p.H + h];
#endif
    const uint32_t a = na, b = nb;
    const float as = lrelu02(nus);
    if (SHADOW_GAT_ROW_PREFETCH && r + rw_.step < rw_.end) { na = p.indptr[r + rw_.step]; nb = p.indptr[r + rw_.step + 1]; nus = p.u_s[(r + rw_.step) * p.H + h]; }
    float mx = -INFINITY;
    uint32_t q = a;
    float den = 0.f;
    float4 acc = make_float4(0, 0, 0, 0);
#if SHADOW_GAT_ONLINE_SOFTMAX
#define SHD_CALL(G) gat_fwd_group_online<G>(p, q, h, f, on, as, mx, den, acc)
    SHD_GAT_EDGES(SHADOW_GAT_GROUPS_FWD, q, b, SHD_CALL);
#undef SHD_CALL
    if (a == b) mx = 0.f;
#else
#define SHD_CALL(G) gat_max_group<G>(p, q, h, as, mx)
    SHD_GAT_EDGES(SHADOW_GAT_GROUPS_FWD, q, b, SHD_CALL);
#undef SHD_CALL
    if (a == b) mx = 0.f;
    q = a;
#define SHD_CALL(G) gat_fwd_group<G>(p, q, h, f, on, as, mx, den, acc)
    SHD_GAT_EDGES(SHADOW_GAT_GROUPS_FWD, q, b, SHD_CALL);
#undef SHD_CALL
#endif
    den = fmaxf(den, 1e-10f);
    const float inv = 1.0f / den;
    if (on) {
      gst4(p.nagg + r * p.F + f, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
      if ((l % ls) == 0) { p.mx[r * p.H + h] = mx; p.den[r * p.H + h] = den; }
    }
  }
}

// backward, row side: alpha / de per edge, du_s, attention part of dz_self, datt[0]
#ifndef SHADOW_GAT_ROW_BWD_WAVES     // (scripts/micro/ab_gat_row_waves.sh: a register cap for more resident wavefronts)
#define SHADOW_GAT_ROW_BWD_ATTR
#else
#define SHADOW_GAT_ROW_BWD_ATTR __attribute__((amdgpu_waves_per_eu(SHADOW_GAT_ROW_BWD_WAVES, 8)))
#endif
template <int LPR>
__global__ void SHADOW_GAT_ROW_BWD_ATTR gat_row_bwd_kernel(GatParams p) {
  const uint32_t rpb = kGatBlock / LPR, sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t f = l * 4, ls = p.D / 4;
  const bool on = f < p.F;
  const uint32_t h = on ? f / p.D : 0;
  float4 a0 = make_float4(0, 0, 0, 0), g0 = a0;
  if (on) a0 = gld4(p.att + f);
  const RowWalk rw_ = xcd_row_walk(p.n, rpb, sub);
  uint32_t na = 0, nb = 0;                              // (next row's pointers ahead: see gat_row_fwd_kernel)
  float4 ndn = make_float4(0, 0, 0, 0), nng = ndn;
  if (rw_.r < rw_.end) {
    na = p.indptr[rw_.r]; nb = p.indptr[rw_.r + 1];
    if (on) { ndn = gld4(p.dnagg + rw_.r * p.F + f); nng = gld4(p.nagg + rw_.r * p.F + f); }
  }
  for (uint64_t r = rw_.r; r < rw_.end; r += rw_.step) {
#if !SHADOW_GAT_ROW_PREFETCH
    na = p.indptr[r]; nb = p.indptr[r + 1];
    if (on) { ndn = gld4(p.dnagg + r * p.F + f); nng = gld4(p.nagg + r * p.F + f); }
#endif
    const uint32_t a = na, b = nb;
    const float4 dn = ndn, ng = nng;
    if (SHADOW_GAT_ROW_PREFETCH && r + rw_.step < rw_.end) {
      const uint64_t r2 = r + rw_.step;
      na = p.indptr[r2]; nb = p.indptr[r2 + 1];
      if (on) { ndn = gld4(p.dnagg + r2 * p.F + f); nng = gld4(p.nagg + r2 * p.F + f); }
    }
    const float t = slice_sum(dot4(dn, ng), ls);          // dN_i . N_i per head
    const float usr = p.u_s[r * p.H + h];
    const float as = lrelu02(usr);
    const float mx = p.mx[r * p.H + h], den_r = p.den[r * p.H + h], inv = 1.0f / den_r;
    float das = 0.f;
    uint32_t q = a;
    const bool lead = on && (l % ls) == 0;
#define SHD_CALL(G) gat_bwd_row_group<G>(p, q, h, f, on, lead, ls, as, mx, inv, t, dn, das)
    SHD_GAT_EDGES(SHADOW_GAT_GROUPS_ROW, q, b, SHD_CALL);
#undef SHD_CALL
    // The score of edge (i, j) is lrelu(u_s[i]) + lrelu(u_n[j]): the row's softmax does not change when u_s[i] moves, so the
    // aggregate's gradient with respect to u_s[i] is EXACTLY zero (sum_j de_ij = t - t sum_j alpha_ij, sum_j alpha_ij = 1) --
    // the reference's autograd produces rounding noise around 0 there (layers.py:568-581).  The one exception is a row whose
    // denominator sits on its 1e-10 clamp (every kept edge more than e^-23 below the dropped maximum): there alpha does not
    // sum to one and the sum is kept.  Everywhere else the attention's share of dz_self and of datt[0] is zero: the lanes of an
    // unclamped head neither read z_self nor touch dz_self (0.9 GB less per launch at 294 k rows).
    const bool clamped = !(den_r > 1e-10f);          // (the forward pass stored max(sum, 1e-10))
    const float dus = clamped ? das * dlrelu02(usr) : 0.f;
    float rmax = 0.f;
    if (on) {
      if ((l % ls) == 0) p.du_s[r * p.H + h] = dus;
      if (clamped) {
        const float4 z = gld4(p.z_self + r * p.F + f);
        const float4 hs = act4(p.act, z);
        float4 dzv = make_float4(dus * a0.x * g_act_bwd(p.act, z.x, hs.x), dus * a0.y * g_act_bwd(p.act, z.y, hs.y),
                                 dus * a0.z * g_act_bwd(p.act, z.z, hs.z), dus * a0.w * g_act_bwd(p.act, z.w, hs.w));
        if (p.acc_self) {                  // (z_self also feeds the layer's act_norm: its gradient share is here already)
          const float4 o = gld4(p.dz_self + r * p.F + f);
          dzv.x += o.x; dzv.y += o.y; dzv.z += o.z; dzv.w += o.w;
        }
        gst4(p.dz_self + r * p.F + f, dzv);
        rmax = shadow::amax4(dzv);
        g0.x += dus * hs.x; g0.y += dus * hs.y; g0.z += dus * hs.z; g0.w += dus * hs.w;
      } else if (!p.acc_self) {
        gst4(p.dz_self + r * p.F + f, make_float4(0.f, 0.f, 0.f, 0.f));
      }
    }
    if (p.row_amax) {                    // (the LPR lanes of a row group share r)
      // accumulate mode: the array holds max |dz_self| of the incoming rows (the caller's contract); what this pass changed
      // is joined in.  Otherwise this pass wrote the whole row.
      rmax = shadow::group_max<LPR>(rmax);
      if (l == 0) {
        if (!p.acc_self) p.row_amax[r] = rmax;
        else if (rmax > 0.f) p.row_amax[r] = fmaxf(p.row_amax[r], rmax);
      }
    }
  }
  // datt[0] += sum over this block's rows
  __shared__ float red[kGatBlock * 4];
  red[threadIdx.x * 4 + 0] = g0.x; red[threadIdx.x * 4 + 1] = g0.y; red[threadIdx.x * 4 + 2] = g0.z; red[threadIdx.x * 4 + 3] = g0.w;
  __syncthreads();
  if (sub == 0 && on) {
    float s4[4] = {0, 0, 0, 0};
    for (uint32_t q = 0; q < rpb; q++)
      for (int k = 0; k < 4; k++) s4[k] += red[(q * LPR + l) * 4 + k];
    for (int k = 0; k < 4; k++) p.datt_part[((size_t)blockIdx.x * 2 + 0) * p.F + f + k] = s4[k];
  }
}

// backward, column side (transposed CSR): d hn, du_n, dz_neigh, datt[1]
template <int LPR>
__global__ void gat_col_bwd_kernel(GatParams p) {
  const uint32_t rpb = kGatBlock / LPR, sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t f = l * 4, ls = p.D / 4;
  const bool on = f < p.F;
  const uint32_t h = on ? f / p.D : 0;
  float4 a1 = make_float4(0, 0, 0, 0), g1 = a1;
  if (on) a1 = gld4(p.att + p.F + f);
  const RowWalk rw_ = xcd_row_walk(p.n, rpb, sub);
  uint32_t na = 0, nb = 0;                              // (next row's pointers ahead: see gat_row_fwd_kernel)
  if (rw_.r < rw_.end) { na = p.t_indptr[rw_.r]; nb = p.t_indptr[rw_.r + 1]; }
  for (uint64_t r = rw_.r; r < rw_.end; r += rw_.step) {
#if !SHADOW_GAT_ROW_PREFETCH
    na = p.t_indptr[r]; nb = p.t_indptr[r + 1];
#endif
    const uint32_t a = na, b = nb;
    if (SHADOW_GAT_ROW_PREFETCH && r + rw_.step < rw_.end) { na = p.t_indptr[r + rw_.step]; nb = p.t_indptr[r + rw_.step + 1]; }
    float4 acc = make_float4(0, 0, 0, 0);
    float dan = 0.f, rmax = 0.f;
    uint32_t q = a;
#define SHD_CALL(G) gat_bwd_col_group<G>(p, q, h, f, on, dan, acc)
    SHD_GAT_EDGES(SHADOW_GAT_GROUPS_COL, q, b, SHD_CALL);
#undef SHD_CALL
    if (on) {
      const float dun = dan * dlrelu02(p.u_n[r * p.H + h]);
      float4 hn, dzv;
      acc.x += dun * a1.x; acc.y += dun * a1.y; acc.z += dun * a1.z; acc.w += dun * a1.w;
      if (p.z_neigh) {
        const float4 z = gld4(p.z_neigh + r * p.F + f);
        hn = p.hn ? gld4(p.hn + r * p.F + f) : act4(p.act, z);
        dzv = make_float4(acc.x * g_act_bwd(p.act, z.x, hn.x), acc.y * g_act_bwd(p.act, z.y, hn.y),
                          acc.z * g_act_bwd(p.act, z.z, hn.z), acc.w * g_act_bwd(p.act, z.w, hn.w));
      } else {                           // (the paired Linear wrote hn = act(z_neigh) straight away: the derivative from hn, gat_act.h)
        hn = gld4(p.hn + r * p.F + f);
        dzv = make_float4(acc.x * g_act_bwd_h(p.act, hn.x), acc.y * g_act_bwd_h(p.act, hn.y),
                          acc.z * g_act_bwd_h(p.act, hn.z), acc.w * g_act_bwd_h(p.act, hn.w));
      }
      gst4(p.dz_neigh + r * p.F + f, dzv);
      rmax = shadow::amax4(dzv);
      g1.x += dun * hn.x; g1.y += dun * hn.y; g1.z += dun * hn.z; g1.w += dun * hn.w;
    }
    if (p.row_amax) {                    // joined with the maximum the row pass left for dz_self's row
      rmax = shadow::group_max<LPR>(rmax);
      if (l == 0) p.row_amax[r] = fmaxf(p.row_amax[r], rmax);
    }
  }
  __shared__ float red[kGatBlock * 4];
  red[threadIdx.x * 4 + 0] = g1.x; red[threadIdx.x * 4 + 1] = g1.y; red[threadIdx.x * 4 + 2] = g1.z; red[threadIdx.x * 4 + 3] = g1.w;
  __syncthreads();
  if (sub == 0 && on) {
    float s4[4] = {0, 0, 0, 0};
    for (uint32_t q = 0; q < rpb; q++)
      for (int k = 0; k < 4; k++) s4[k] += red[(q * LPR + l) * 4 + k];
    for (int k = 0; k < 4; k++) p.datt_part[((size_t)blockIdx.x * 2 + 1) * p.F + f + k] = s4[k];
  }
}

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
  SHD_GAT_LAUNCH(gat_row_fwd_kernel, lpr, g, st, p);
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
  SHD_GAT_LAUNCH(gat_row_fwd_kernel, lpr, g, st, p);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sl_gat_bwd(const uint32_t *d_indptr, const uint32_t *d_indices, const uint32_t *d_t_indptr,
                          const uint32_t *d_t_indices, const uint32_t *d_t_perm, const float *d_edge_w,
                          const float *d_z_self, const float *d_z_neigh, const float *d_att, int act,
                          uint32_t n, uint32_t e, uint32_t F, uint32_t heads, const float *d_hn,
                          const float *d_u_s, const float *d_u_n, const float *d_mx, const float *d_den,
                          const float *d_nagg, const float *d_dnagg, float *d_work, float *d_dz_self,
                          float *d_dz_neigh, float *d_datt, int accumulate_dz_self, float *d_row_amax, void *stream_) {
  if (!d_indptr || !d_t_indptr || !d_z_self || !(d_z_neigh || d_hn) || !d_att || !d_u_s || !d_u_n || !d_mx ||
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
  p.dnagg = d_dnagg;
  // work: alpha[e*H], de[e*H], du_s[n*H], datt_part[2048][2][F]
  p.alpha = d_work; p.de = d_work + (size_t)e * heads; p.du_s = p.de + (size_t)e * heads;
  p.datt_part = p.du_s + (size_t)n * heads;
  p.dz_self = d_dz_self; p.dz_neigh = d_dz_neigh; p.datt = d_datt; p.acc_self = accumulate_dz_self; p.row_amax = d_row_amax;
  const uint32_t g = gat_grid(n, lpr);
  SHD_GAT_LAUNCH(gat_row_bwd_kernel, lpr, g, st, p);
  SHD_GAT_LAUNCH(gat_col_bwd_kernel, lpr, g, st, p);
  hipLaunchKernelGGL(gat_datt_finish_kernel, dim3((2 * F + kDattCols - 1) / kDattCols), dim3(kDattCols * kDattSlices), 0, st,
                     p.datt_part, g, 2 * F, d_datt);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}
