// sampler_scan.h -- device side of the MI355X subgraph sampler, part 2: the node-induced slicing
// (ParallelSampler::_node_induced_subgraph, ParallelSampler.cpp:378-431) as a flat, chip-wide scan.
//
// The selection kernel (sampler_device.h) leaves, per subgraph, the sorted node list and one 16-byte record per
// row {e0, deg, slot prefix, id} plus the prefix of the rows' aligned 16-byte QUADS in the full graph's indices
// array.  The concatenated quads of all rows of all subgraphs are the stream this kernel reads exactly once:
//
//   sg_plan_kernel   (one workgroup) cuts every subgraph's quad range into work ITEMS of `cpi` chunks of kQChunk
//                    quads (cpi grows with the batch so that the item count stays bounded) and writes the item
//                    prefix.
//   sg_scan_kernel   persistent workgroups pull items from a global ticket -- no workgroup is tied to a subgraph,
//                    so every CU streams ids for the whole duration and one heavy subgraph spreads over the chip.
//                    Per item: rebuild the subgraph's membership filter in LDS (bit = id mod 2^k; a clear bit is a
//                    definite miss), then every WAVEFRONT walks one chunk ROW BY ROW with wave-uniform row scalars:
//                    a row's quads are covered by coalesced 16-byte-per-lane loads (<= 64 quads = 1 KiB per
//                    instruction, all of a chunk's loads in flight before the first is consumed), each id costs one
//                    LDS dword probe, and the rare candidates (~1 % of the ids) go to an LDS list keyed by
//                    2*slot+kind.  At the end of the round the candidates are resolved exactly (binary search in
//                    the sorted node list = the sub id), bucket-sorted by key -- which restores the reference's
//                    edge order -- and appended to the subgraph's edge scratch; a round record remembers where.
//                    Key ranges of different items / rounds of a subgraph are disjoint and ordered by quad position,
//                    so the relocation kernel only concatenates the records in item order.
//
// Self-edge insertion (.cpp:386-400), the reference's over-read (compat) and the root<->root exclusion of
// multi-root subgraphs (.cpp:414-418) ride along: the lane that holds a row's last neighbour decides the trailing
// self edge, rows without neighbours are handled by the wavefront that finishes the preceding row.
#pragma once
#include "sampler_device.h"

namespace shadow {

constexpr uint32_t kMaxRoundChunks = 64;     // chunks a round may span (start-row table in LDS)
constexpr uint32_t kPlanItems = 8192;        // item budget the plan aims at (beyond one item per subgraph)

// ---------------------------------------------------------------------------------------------------------------
// plan: item prefix over the subgraphs
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) sg_plan_kernel(SampleParams p) {
  __shared__ uint32_t wsum[32];
  __shared__ uint32_t s_cpi;
  const uint32_t tid = threadIdx.x, T = blockDim.x;
  uint32_t chunks = 0;
  for (uint32_t s = tid; s < p.P; s += T) {
    const uint32_t *c = p.s_cnt + (size_t)s * R_WORDS;
    if (!(c[R_FLAGS] & 1u)) chunks += (c[R_Q] + kQChunk - 1u) / kQChunk;
  }
  uint32_t total;
  (void)block_excl_scan(chunks, wsum, &total);
  if (tid == 0) {
    uint32_t cpi = 16;
    const uint32_t need = (total + kPlanItems - 1u) / kPlanItems;
    if (need > cpi) cpi = need;
    s_cpi = cpi;
  }
  __syncthreads();
  const uint32_t cpi = s_cpi;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < p.P; base += T) {
    const uint32_t s = base + tid;
    uint32_t items = 0;
    if (s < p.P) {
      const uint32_t *c = p.s_cnt + (size_t)s * R_WORDS;
      if (!(c[R_FLAGS] & 1u)) {
        const uint32_t ch = (c[R_Q] + kQChunk - 1u) / kQChunk;
        items = ch ? (ch + cpi - 1u) / cpi : 1u;         // a subgraph without neighbours still owns one item
      }
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan(items, wsum, &tot);
    if (s < p.P) p.itemptr[s] = carry + ex;
    carry += tot;
  }
  if (tid == 0) {
    p.itemptr[p.P] = carry;
    p.plan[PL_NITEMS] = carry; p.plan[PL_CPI] = cpi; p.plan[PL_POOL] = carry; p.plan[PL_FLAGS] = 0;
    p.plan[PL_TICKET] = 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// scan
// ---------------------------------------------------------------------------------------------------------------
struct ScanLds {
  uint32_t *bits;    // [bit_words]
  uint32_t *lkey;    // [capm] candidate list: 2*slot+kind
  uint32_t *lval;    // [capm] neighbour's global id (kind 1) / unused (kind 0)
  uint32_t *lrow;    // [capm] row of the entry
  uint32_t *lnext;   // [capm] bucket chains of the final sort
  uint32_t *bhead;   // [kSortBuckets]
  uint32_t *bcnt;    // [kSortBuckets]
  uint32_t *nodes;   // [nodes_lds] sorted node ids of the subgraph (when they fit)
  uint32_t *crow;    // [kMaxRoundChunks] first row of every chunk of the round
  unsigned char *wtmp;  // [waves * 64] wave-private start-position flags
  uint32_t *ctrl;    // [C_WORDS]
};

struct ScanLayout {
  size_t bits, lkey, lval, lrow, lnext, bhead, bcnt, nodes, crow, wtmp, ctrl, total;
};

__host__ __device__ inline ScanLayout scan_layout(uint32_t bit_words, uint32_t capm, uint32_t nodes_lds) {
  ScanLayout L;
  size_t o = 0;
  L.bits = o; o += (size_t)bit_words * 4;
  L.lkey = o; o += r16((size_t)capm * 4);
  L.lval = o; o += r16((size_t)capm * 4);
  L.lrow = o; o += r16((size_t)capm * 4);
  L.lnext = o; o += r16((size_t)capm * 4);
  L.bhead = o; o += kSortBuckets * 4;
  L.bcnt = o; o += kSortBuckets * 4;
  L.nodes = o; o += r16((size_t)nodes_lds * 4);
  L.crow = o; o += kMaxRoundChunks * 4;
  L.wtmp = o; o += 16 * 64;
  L.ctrl = o; o += C_WORDS * 4;
  L.total = o;
  return L;
}

// one list entry per set bit of `mask` (bit q -> key keys[q], value vals[q]); per-lane LDS append
__device__ __forceinline__ void list_put(const ScanLds &t, uint32_t capm, uint32_t &r, uint32_t key, uint32_t val,
                                         uint32_t row) {
  if (r < capm) { t.lkey[r] = key; t.lval[r] = val; t.lrow[r] = row; }
  r++;
}

// wave-uniform 64-way search: largest s in [0, P) with itemptr[s] <= item (itemptr non-decreasing, itemptr[P] > item)
__device__ __forceinline__ uint32_t find_subgraph(const uint32_t *itemptr, uint32_t P, uint32_t item) {
  uint32_t lo = 0, hi = P;          // answer in [lo, hi)
  const uint32_t lane = lane_id();
  while (hi - lo > 1) {
    const uint32_t span = hi - lo;
    const uint32_t step = (span + 63u) / 64u;
    const uint32_t probe = lo + min(span, (lane + 1u) * step);          // candidates lo+step, lo+2 step, ... (<= hi)
    const bool le = (probe < hi) && (itemptr[probe] <= item);
    const uint64_t m = __ballot(le);
    const uint32_t k = (uint32_t)__popcll(m);                          // monotone: the first k probes are <= item
    const uint32_t nlo = lo + min(span, k * step);
    const uint32_t nhi = min(hi, lo + (k + 1u) * step);
    lo = (k == 0) ? lo : nlo;
    hi = nhi;
  }
  return lo;
}

// 64 consecutive rows of a subgraph held one per lane (the scan's row WINDOW): the lanes' quad counts and their
// exclusive prefix turn a window-local quad position into its row with one byte scatter + a fill-forward scan.
struct RowWin {
  uint32_t e0, deg, rs, v;   // this lane's row (zeros beyond the last row)
  uint32_t nq;               // aligned quads of the row
  uint32_t wq;               // quads of the window's preceding rows
};

__device__ __forceinline__ uint4 win_fetch(const RowInfo *info, uint32_t wb, uint32_t n) {
  const uint32_t r = wb + lane_id();
  return (r < n) ? *reinterpret_cast<const uint4 *>(info + r) : make_uint4(0u, 0u, 0u, 0u);
}

__device__ __forceinline__ RowWin win_make(const uint4 w, uint32_t *total) {
  RowWin r;
  r.e0 = w.x; r.deg = w.y; r.rs = w.z; r.v = w.w;
  r.nq = w.y ? (((w.x + w.y - 1u) >> 2) - (w.x >> 2) + 1u) : 0u;
  const uint32_t incl = wave_incl_scan(r.nq);
  r.wq = incl - r.nq;
  *total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  return r;
}

__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t src_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

constexpr int kBatch = 4;                    // row-packed groups of 64 quads (1 KiB) a wavefront keeps in flight

// sentinel-slot entries of a row WITHOUT neighbours (deg = 0): the self edge (.cpp:387-400 with an empty row) and,
// in compat mode without self edges, the over-read of the element at the row's (empty) position (.cpp:401-405)
template <bool kPlain>
__device__ __forceinline__ void emit_empty_row(const SampleParams &p, const ScanLds &t, uint32_t *ctrl, const RowInfo *info,
                                               uint32_t r, bool incl_self, bool compat, uint32_t bw_mask, uint32_t capm) {
  const uint4 w = *reinterpret_cast<const uint4 *>(info + r);
  uint32_t over = 0;
  bool put_over = false;
  uint32_t cnt = incl_self ? 1u : 0u;
  if (compat && !incl_self && (uint64_t)w.x < p.nnz) {
    over = p.indices[w.x];
    put_over = ((t.bits[(over >> 5) & bw_mask] >> (over & 31u)) & 1u) != 0;
    if (put_over) cnt++;
  }
  if (cnt) {
    uint32_t rr = atomicAdd(&ctrl[C_M], cnt);
    if (incl_self) list_put(t, capm, rr, 2u * w.z, 0u, r);
    if (put_over) list_put(t, capm, rr, 2u * w.z + 1u, over, r);
  }
}

template <bool kPlain>
__global__ void sg_scan_kernel(SampleParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ScanLayout L = scan_layout(p.bit_words, p.capm, p.nodes_lds);
  ScanLds t;
  t.bits = (uint32_t *)(smem + L.bits);
  t.lkey = (uint32_t *)(smem + L.lkey);
  t.lval = (uint32_t *)(smem + L.lval);
  t.lrow = (uint32_t *)(smem + L.lrow);
  t.lnext = (uint32_t *)(smem + L.lnext);
  t.bhead = (uint32_t *)(smem + L.bhead);
  t.bcnt = (uint32_t *)(smem + L.bcnt);
  t.nodes = (uint32_t *)(smem + L.nodes);
  t.crow = (uint32_t *)(smem + L.crow);
  t.wtmp = smem + L.wtmp;
  uint32_t *ctrl = (uint32_t *)(smem + L.ctrl);
  __shared__ uint32_t s_item;

  const uint32_t tid = threadIdx.x, T = blockDim.x;
  const uint32_t lane = lane_id(), wave = wave_id(), nw = T >> 6;
  const uint32_t bw_mask = p.bit_words - 1u;
  const uint32_t capm = p.capm, cape = p.cap_edges_scr;
  const int R = p.R;
  const bool incl_self = !kPlain && (p.include_self != 0);
  const bool itc = kPlain || (p.include_target_conn != 0) || (R == 1);   // .cpp:356-358
  const bool compat = !kPlain && (p.compat != 0);
  const bool sentinel = incl_self || compat;
  const uint32_t nitems = p.plan[PL_NITEMS], cpi = p.plan[PL_CPI];
  unsigned char *wflag = t.wtmp + wave * 64u;            // wave-private, all zero between uses
  for (uint32_t i = tid; i < 16u * 16u; i += T) reinterpret_cast<uint32_t *>(t.wtmp)[i] = 0;

  for (;;) {
    if (tid == 0) s_item = atomicAdd(&p.plan[PL_TICKET], 1u);
    __syncthreads();
    const uint32_t item = s_item;
    __syncthreads();
    if (item >= nitems) return;
    const uint32_t s = find_subgraph(p.itemptr, p.P, item);
    const uint32_t li = item - p.itemptr[s];
    uint32_t *res = p.s_cnt + (size_t)s * R_WORDS;
    const uint32_t n = res[R_N], Q = res[R_Q];
    const uint32_t *roots = p.roots + (size_t)s * R;
    const uint32_t *g_nodes = p.s_nodes + (size_t)s * p.cap_nodes_scr;
    const RowInfo *g_info = p.s_rowinfo + (size_t)s * p.cap_nodes_scr;
    const uint32_t *g_rowq = p.s_rowq + (size_t)s * (p.cap_nodes_scr + 1);
    uint32_t *g_row = p.s_row + (size_t)s * cape;
    uint32_t *g_col = p.s_col + (size_t)s * cape;
    uint32_t *g_eid = p.s_eid + (size_t)s * cape;
    const bool nodes_in_lds = n <= p.nodes_lds;
    const uint32_t iq0 = min(Q, li * cpi * kQChunk);
    const uint32_t iq1 = min(Q, iq0 + cpi * kQChunk);

    // ---- per item: the subgraph's membership filter (+ its sorted node list when it fits)
    for (uint32_t w = tid; w < p.bit_words; w += T) t.bits[w] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += T) {
      const uint32_t v = g_nodes[i];
      atomicOr(&t.bits[(v >> 5) & bw_mask], 1u << (v & 31u));
      if (nodes_in_lds) t.nodes[i] = v;
    }
    // (the first barrier of the round loop orders these writes before the scan)

    uint32_t rec_prev = 0xFFFFFFFFu;           // last round record of this item (thread 0)
    uint32_t rq0 = iq0;
    uint32_t rquads = min(iq1 - iq0, kMaxRoundChunks * kQChunk);
    bool first_round = true;
    for (;;) {
      const uint32_t rq1 = min(iq1, rq0 + max(rquads, 1u));
      const uint32_t rchunks = (rq1 - rq0 + kQChunk - 1u) / kQChunk;
      if (tid == 0) ctrl[C_M] = 0;
      // ---- first row of every chunk of the round: the row that holds the chunk's first quad
      for (uint32_t base = 0; base < n; base += T) {
        const uint32_t r = base + tid;
        if (r < n) {
          const uint32_t qs = g_rowq[r], qe = g_rowq[r + 1];
          if (qe > qs && qe > rq0 && qs < rq1) {
            uint32_t k = qs > rq0 ? (qs - rq0 + kQChunk - 1u) / kQChunk : 0u;
            for (; k < rchunks && rq0 + k * kQChunk < qe; k++) t.crow[k] = r;
          }
        }
      }
      __syncthreads();

      // ---- the scan: wave w takes chunks w, w + nw, ... of the round
      for (uint32_t k = wave; k < max(rchunks, 1u); k += nw) {
        const uint32_t qa = rq0 + k * kQChunk, qb = min(qa + kQChunk, rq1);
        // rows without neighbours in front of the subgraph's first quad (or a subgraph without any quad):
        // their sentinel slots belong to the very first chunk of the subgraph
        if (!kPlain && sentinel && qa == 0 && k == 0 && li == 0) {
          const uint32_t stop = (Q == 0) ? n : rl_first(t.crow[0]);
          for (uint32_t r0 = 0; r0 < stop; r0 += 64)
            if (r0 + lane < stop) emit_empty_row<kPlain>(p, t, ctrl, g_info, r0 + lane, incl_self, compat, bw_mask, capm);
        }
        if (rchunks == 0) break;
        // ---- window of 64 rows starting at the row that holds the chunk's first quad
        uint32_t wb = rl_first(t.crow[k]);
        uint32_t W;
        RowWin win = win_make(win_fetch(g_info, wb, n), &W);
        uint4 wnext = win_fetch(g_info, wb + 64u, n);                  // prefetched
        uint32_t P0 = qa - rl_first(g_rowq[wb]);                       // window-local position of the next quad
        uint32_t qpos = qa;
        bool at_window_end = false;
        while (qpos < qb) {
          uint4 c4[kBatch];
          uint32_t m_j0[kBatch], m_deg[kBatch], m_rs[kBatch], m_row[kBatch], m_v[kBatch], m_prev[kBatch], m_act[kBatch];
          uint32_t m_e0[kBatch], m_take[kBatch];
#pragma unroll
          for (int u = 0; u < kBatch; u++) {
            m_act[u] = 0; m_j0[u] = 0; m_deg[u] = 0; m_rs[u] = 0; m_row[u] = 0; m_v[u] = 0; m_prev[u] = 0;
            m_e0[u] = 0; m_take[u] = 0;
            c4[u] = make_uint4(kEmpty, kEmpty, kEmpty, kEmpty);
            if (qpos < qb) {
              if (P0 >= W) {
                // window consumed: the next 64 rows.  Rows without neighbours at the head of the new window sit at
                // the position the previous group ended on: their sentinel slots are this wavefront's.  (A window
                // can consist of such rows only: W == 0 -> keep advancing.)
                do {
                  wb += 64u;
                  win = win_make(wnext, &W);
                  wnext = win_fetch(g_info, wb + 64u, n);
                  if (!kPlain && sentinel && wb + lane < n && win.nq == 0 && win.wq == 0)
                    emit_empty_row<kPlain>(p, t, ctrl, g_info, wb + lane, incl_self, compat, bw_mask, capm);
                } while (W == 0 && wb + 64u < n);
                P0 = 0;
              }
              const uint32_t take = min(min(64u, W - P0), qb - qpos);
              // position -> row: rows that start inside (P0, P0 + take) flag their start position; fill forward
              const bool nonempty = win.nq != 0;
              const uint64_t before = __ballot(nonempty && win.wq <= P0);
              const uint32_t f0 = 63u - (uint32_t)__builtin_clzll((unsigned long long)before);   // row that holds P0
              const bool starts = nonempty && win.wq > P0 && win.wq < P0 + take;
              if (starts) wflag[win.wq - P0] = (unsigned char)(lane + 1u);
              __builtin_amdgcn_wave_barrier();
              uint32_t rl = wflag[lane];
              __builtin_amdgcn_wave_barrier();
              if (starts) wflag[win.wq - P0] = 0;
              if (lane == 0) rl = f0 + 1u;
#pragma unroll
              for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)rl, off, 64);
                if (lane >= (uint32_t)off) rl = max(rl, o);
              }
              rl -= 1u;                                                     // lane of the window that owns my position
              const uint32_t e0 = lane_get(win.e0, rl), wq = lane_get(win.wq, rl);
              const uint32_t deg = lane_get(win.deg, rl), rs = lane_get(win.rs, rl);
              const uint32_t kq = P0 + lane - wq;                            // quad index inside the row
              const uint32_t addr = ((e0 >> 2) + kq) << 2;
              const bool act = lane < take;
              m_act[u] = act ? 1u : 0u;
              m_j0[u] = addr - e0;                                          // wraps when the quad starts before the row
              m_deg[u] = deg; m_rs[u] = rs; m_row[u] = wb + rl;
              if (act) c4[u] = *reinterpret_cast<const uint4 *>(p.indices + addr);
              if (!kPlain) {
                m_e0[u] = e0;
                m_v[u] = lane_get(win.v, rl);
                if (incl_self && lane == 0 && kq > 0) m_prev[u] = p.indices[addr - 1u];   // previous quad of the row is not in this group
                // rows without neighbours whose position lies in (P0, P0 + take]: their sentinel slots go with this group
                if (sentinel && wb + lane < n && win.nq == 0 && win.wq > P0 && win.wq <= P0 + take)
                  emit_empty_row<kPlain>(p, t, ctrl, g_info, wb + lane, incl_self, compat, bw_mask, capm);
              }
              m_take[u] = take;
              P0 += take; qpos += take;
              at_window_end = (P0 >= W);
            }
          }
          // ---- probe + emit
#pragma unroll
          for (int u = 0; u < kBatch; u++) {
            if (m_take[u] == 0) continue;                                   // wave-uniform
            const uint32_t deg = m_deg[u], rs = m_rs[u], j0 = m_j0[u];
            const uint32_t cc[4] = {c4[u].x, c4[u].y, c4[u].z, c4[u].w};
            uint32_t vmask = 0;
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (m_act[u] && (j0 + c < deg)) vmask |= 1u << c;
            uint32_t hit = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
              const uint32_t w = t.bits[(cc[c] >> 5) & bw_mask];
              hit |= ((w >> (cc[c] & 31u)) & 1u) << c;
            }
            hit &= vmask;
            uint32_t selfm = 0, trail = 0, last_c = 0;
            bool fin = false;                          // this lane holds the row's last neighbour
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (((vmask >> c) & 1u) && j0 + c + 1u == deg) { fin = true; last_c = cc[c]; }
            if (!kPlain && incl_self) {
              // the self edge goes right before the first neighbour > v (.cpp:387-400, :408-410), or after the last one
              const uint32_t v = m_v[u];
              const uint32_t up = (uint32_t)__shfl_up((int)cc[3], 1, 64);
              const uint32_t pv0 = (lane == 0) ? m_prev[u] : up;
#pragma unroll
              for (int c = 0; c < 4; c++) {
                const uint32_t j = j0 + c;
                const uint32_t pv = (c == 0) ? pv0 : cc[c > 0 ? c - 1 : 0];
                const bool prev_lt = (j == 0) || (pv < v);
                if (((vmask >> c) & 1u) && prev_lt && v < cc[c]) selfm |= 1u << c;
              }
              if (fin && last_c < v) trail = 1;
            }
            const uint32_t cnt = (uint32_t)(__popc(hit) + __popc(selfm)) + trail;
            if (cnt) {
              uint32_t r = atomicAdd(&ctrl[C_M], cnt);
              const uint32_t keybase = 2u * (rs + j0);
#pragma unroll
              for (int c = 0; c < 4; c++) {
                if (!kPlain && ((selfm >> c) & 1u)) list_put(t, capm, r, keybase + 2u * c, 0u, m_row[u]);   // .cpp:408-410
                if ((hit >> c) & 1u) list_put(t, capm, r, keybase + 2u * c + 1u, cc[c], m_row[u]);       // .cpp:420-422
              }
              if (!kPlain && trail) list_put(t, capm, r, 2u * (rs + deg), 0u, m_row[u]);
            }
            if (!kPlain && compat && fin && !trail) {
              // reference over-read (.cpp:401-405): when no self edge was inserted INSIDE the row (and none trails),
              // the element right behind the row is examined like a neighbour
              const uint32_t row_e0 = m_e0[u];
              bool inserted = false;
              if (incl_self) {
                uint32_t l3 = 0, h3 = deg;
                while (l3 < h3) { const uint32_t m3 = (l3 + h3) >> 1; if (p.indices[row_e0 + m3] < m_v[u]) l3 = m3 + 1; else h3 = m3; }
                inserted = !(l3 < deg && p.indices[row_e0 + l3] == m_v[u]);
              }
              if (!inserted && (uint64_t)row_e0 + deg < p.nnz) {
                const uint32_t c = p.indices[row_e0 + deg];
                if ((t.bits[(c >> 5) & bw_mask] >> (c & 31u)) & 1u) {
                  uint32_t r = atomicAdd(&ctrl[C_M], 1u);
                  list_put(t, capm, r, 2u * (rs + deg) + 1u, c, m_row[u]);
                }
              }
            }
          }
        }
        // the chunk's last group ended exactly at the end of its window: rows without neighbours may follow in the
        // next window(s) -- they belong to this chunk
        if (!kPlain && sentinel && at_window_end) {
          uint32_t nb = wb + 64u;
          uint4 w4 = wnext;
          while (nb < n) {
            const uint32_t r2 = nb + lane;
            const uint64_t ne = __ballot(r2 < n && w4.y != 0);
            const uint32_t first_ne = ne ? (uint32_t)__ffsll((unsigned long long)ne) - 1u : 64u;
            if (r2 < n && lane < first_ne) emit_empty_row<kPlain>(p, t, ctrl, g_info, r2, incl_self, compat, bw_mask, capm);
            if (ne) break;
            nb += 64u;
            w4 = win_fetch(g_info, nb, n);
          }
        }
      }
      __syncthreads();
      const uint32_t m = ctrl[C_M];
      if (m > capm) {
        // too many candidates for the list: redo this round on half the quads
        if (rq1 - rq0 <= 64u) {                       // cannot shrink further (the host sizes capm for 64 quads)
          if (tid == 0) atomicOr(&res[R_FLAGS], 2u);
          break;
        }
        rquads = max((rq1 - rq0) / 2, 64u);
        __syncthreads();
        continue;
      }
      // ---- resolve the neighbour candidates exactly (.cpp:412-413), order the survivors by key
      // (the keys of a round cover a narrow range: bucket on the offset from the round's smallest key, so that the
      //  256 buckets spread over the round only)
      for (uint32_t i = tid; i < kSortBuckets; i += T) { t.bhead[i] = kEmpty; t.bcnt[i] = 0; }
      if (tid == 0) { ctrl[C_MV] = 0; ctrl[C_CHANGED] = 0xFFFFFFFFu; ctrl[C_MFAIL] = 0; }
      __syncthreads();
      for (uint32_t i = tid; i < m; i += T) {
        uint32_t key = t.lkey[i];
        if (key & 1u) {
          const uint32_t c = t.lval[i];
          uint32_t lo = 0, hi = n;
          if (nodes_in_lds) {
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (t.nodes[mid] < c) lo = mid + 1; else hi = mid; }
          } else {
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (g_nodes[mid] < c) lo = mid + 1; else hi = mid; }
          }
          bool keep = lo < n && (nodes_in_lds ? t.nodes[lo] : g_nodes[lo]) == c;
          if (keep && !itc && is_root(roots, R, c)) {
            // multi-root subgraph without include_target_conn: drop root<->root edges (.cpp:414-418)
            const uint32_t rr = t.lrow[i];
            keep = !is_root(roots, R, nodes_in_lds ? t.nodes[rr] : g_nodes[rr]);
          }
          if (keep) t.lval[i] = lo;
          else { key = kEmpty; t.lkey[i] = kEmpty; }
        } else {
          t.lval[i] = t.lrow[i];                        // self edge: column = the row itself
        }
        if (key != kEmpty) { atomicMin(&ctrl[C_CHANGED], key); atomicMax(&ctrl[C_MFAIL], key); }
      }
      __syncthreads();
      const uint32_t kmin = ctrl[C_CHANGED], kmax = ctrl[C_MFAIL];
      uint32_t bshift = 0;
      if (kmin != 0xFFFFFFFFu) { while (((kmax - kmin) >> bshift) >= kSortBuckets) bshift++; }
      for (uint32_t i = tid; i < m; i += T) {
        const uint32_t key = t.lkey[i];
        if (key != kEmpty) {
          const uint32_t b = (key - kmin) >> bshift;
          atomicAdd(&t.bcnt[b], 1u);
          t.lnext[i] = atomicExch(&t.bhead[b], i);
        }
      }
      __syncthreads();
      // exclusive scan of the bucket counts (kSortBuckets = 256 = 4 per lane of wave 0)
      if (wave == 0) {
        uint32_t b4[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { b4[q] = t.bcnt[lane * 4 + q]; sum += b4[q]; }
        const uint32_t incl = wave_incl_scan(sum);
        uint32_t run = incl - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) { t.bcnt[lane * 4 + q] = run; run += b4[q]; }
        if (lane == 63) {
          // reserve the survivors' place in the subgraph's edge scratch and file the round record
          const uint32_t mv = run;
          const uint32_t off = atomicAdd(&res[R_E], mv);
          ctrl[C_MV] = mv; ctrl[C_TICKET] = off;
          uint32_t rec = item;
          if (!first_round) {
            rec = atomicAdd(&p.plan[PL_POOL], 1u);
            if (rec >= p.rec_cap) { atomicOr(&p.plan[PL_FLAGS], 16u); rec = 0xFFFFFFFFu; }
          }
          ctrl[C_NF0] = rec;
        }
      }
      __syncthreads();
      const uint32_t mv = ctrl[C_MV], e_base = ctrl[C_TICKET], rec = ctrl[C_NF0];
      if (tid == 0 && rec != 0xFFFFFFFFu) {
        RoundRec rr;
        rr.src_off = e_base; rr.cnt = mv; rr.next = 0xFFFFFFFFu; rr.pad = 0;
        p.recs[rec] = rr;
        if (rec_prev != 0xFFFFFFFFu) p.recs[rec_prev].next = rec;
        rec_prev = rec;
      }
      for (uint32_t i = tid; i < m; i += T) {
        const uint32_t key = t.lkey[i];
        if (key == kEmpty) continue;
        const uint32_t b = (key - kmin) >> bshift;
        uint32_t r = t.bcnt[b];
        for (uint32_t j = t.bhead[b]; j != kEmpty; j = t.lnext[j]) r += (t.lkey[j] < key) ? 1u : 0u;
        const uint32_t o = e_base + r;
        if (o < cape) {
          const uint32_t rw = t.lrow[i];
          g_row[o] = rw;
          g_col[o] = t.lval[i];
          uint32_t eid = 0xFFFFFFFFu;                                             // inserted self edge (.cpp:410)
          if (key & 1u) { const RowInfo ri = g_info[rw]; eid = ri.e0 + ((key >> 1) - ri.rs); }   // .cpp:422
          g_eid[o] = eid;
        }
      }
      first_round = false;
      rq0 = rq1;
      __syncthreads();
      if (rq0 >= iq1) break;
    }
    __syncthreads();
  }
}

}  // namespace shadow
