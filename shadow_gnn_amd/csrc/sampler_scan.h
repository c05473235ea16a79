// sampler_scan.h -- device side of the MI355X subgraph sampler, part 2: the node-induced slicing
// (ParallelSampler::_node_induced_subgraph, ParallelSampler.cpp:378-431) as a flat, chip-wide scan.
//
// The selection kernel (sampler_device.h) leaves, per subgraph, the sorted node list and one 16-byte record per
// row {e0, deg, slot prefix, id} plus the prefix of the rows' aligned 16-byte QUADS in the full graph's indices
// array.  The concatenated quads of all rows of all subgraphs are the stream this kernel reads exactly once:
//
//   sg_plan_kernel        (one workgroup) prefix of the subgraphs' CHUNKS (kQChunk quads = 8 KB of ids, plus seg_pad virtual
//                         chunks per subgraph that stand for its fixed work); the global chunk sequence is cut into equal
//                         spans, one per scan workgroup: balanced in bytes and in per-subgraph overhead, no work queue,
//                         and a heavy subgraph simply spreads over several workgroups.
//   sg_scan_plain_kernel  (plain calls: no self-edge insertion, no compat over-read, single root or include_target_conn)
//   sg_scan_kernel        (everything else)
//                         workgroup w owns chunks [w * cpw, (w + 1) * cpw) -- usually the tail of one subgraph, a few
//                         whole ones and the head of another.  Per subgraph segment: rebuild that subgraph's membership
//                         filter in LDS (bit = id mod 2^k; a clear bit is a definite miss), then ROUNDS of <= 64 chunks in
//                         which the wavefronts stream the quads with coalesced 16-byte-per-lane loads (64 quads = 1 KiB
//                         per instruction, eight in flight), probe every id against the filter (one LDS dword) and note
//                         the rare candidates (~1 % of the ids) in an LDS list.  The two kernels differ in how a round's
//                         quads reach the lanes: a flat list of RUNS (<= 64 quads of one row) built per round and a
//                         software-pipelined loop over it (plain), or row WINDOWS (64 rows held one per lane: long rows as
//                         runs, the quads of short rows packed into shared instructions; position -> row by one byte
//                         scatter and a DPP fill-forward).
//   finish_round          end of a round, shared: candidates resolved exactly (binary search in the sorted node list =
//                         the sub id), counting-sorted by key 2*slot+kind -- which restores the reference's edge order --
//                         and appended to the subgraph's edge scratch; a round record remembers where.  Key ranges of
//                         different rounds / workgroups of a subgraph are disjoint and ordered by quad position, so the
//                         relocation kernel only concatenates the records in workgroup order.
//
// Self-edge insertion (.cpp:386-400), the reference's over-read (compat) and the root<->root exclusion of
// multi-root subgraphs (.cpp:414-418) live in sg_scan_kernel: the lane that holds a row's last neighbour decides the
// trailing self edge, rows without neighbours are handled by the wavefront that finishes the preceding row.
#pragma once
#include "sampler_device.h"

namespace shadow {

constexpr uint32_t kMaxRoundChunks = 64;     // chunks a round may span (start-row table in LDS)
constexpr uint32_t kRecPerBlock = 32;        // round records per block (every scan workgroup owns block w; more come from a pool)
constexpr uint32_t kRankShift = 20;          // rank of a list entry rides in the upper bits of its row word

// ---------------------------------------------------------------------------------------------------------------
// plan: chunk prefix over the subgraphs, chunks per scan workgroup
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) sg_plan_kernel(SampleParams p) {
  __shared__ uint32_t wsum[32];
  const uint32_t tid = threadIdx.x, T = blockDim.x;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < p.P; base += T) {
    const uint32_t s = base + tid;
    uint32_t ch = 0;
    if (s < p.P) {
      const uint32_t *c = p.s_cnt + (size_t)s * R_WORDS;
      // (a subgraph without any neighbour still owns one chunk: its rows' sentinel slots are handled there)
      // + seg_pad virtual chunks in front: the span partition then balances bytes AND the per-segment fixed work
      // (filter build, run list, candidate sort -- worth about as much as streaming two dozen chunks)
      if (!(c[R_FLAGS] & 1u)) ch = max(1u, (c[R_Q] + kQChunk - 1u) / kQChunk) + p.seg_pad;
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan(ch, wsum, &tot);
    if (s < p.P) p.cstart[s] = carry + ex;
    carry += tot;
  }
  if (tid == 0) {
    p.cstart[p.P] = carry;
    p.plan[PL_NCHUNKS] = carry;
    p.plan[PL_CPW] = max(1u, (carry + p.scan_grid - 1u) / p.scan_grid);
    p.plan[PL_POOL] = p.scan_grid;                         // blocks [0, scan_grid) belong to the workgroups
    p.plan[PL_FLAGS] = 0;
    for (int i = PL_T0; i < PL_T0 + 12; i++) p.plan[i] = 0;      // (words 20..22: the selection kernel's phase cycles of this call)
  }
}

// ---------------------------------------------------------------------------------------------------------------
// scan
// ---------------------------------------------------------------------------------------------------------------
struct ScanLds {
  uint32_t *bits;    // [bit_words]
  uint2 *lkn;        // [capm] candidate list: {key = 2*slot+kind, next entry of its bucket chain in the final sort} (one ds_read_b64 per chain step)
  uint32_t *lval;    // [capm] position of the neighbour in `indices` (kind 1: its id is fetched when the round resolves), later the column sub id
  uint32_t *lrow;    // [capm] row of the entry (+ its rank in the upper bits during the write-out)
  uint32_t *bhead;   // [kSortBuckets]
  uint32_t *bcnt;    // [kSortBuckets]
  uint32_t *nodes;   // [nodes_lds] sorted node ids of the subgraph (when they fit)
  uint32_t *crow;    // [kMaxRoundChunks] first row of every chunk of the round
  unsigned char *wtmp;  // [waves * 512] wave-private: start-position flags + records of the runs in flight
  uint32_t *ctrl;    // [C_WORDS]
};

constexpr uint32_t kWavePriv = 640;          // wave-private LDS: 64 start-position flags + 36 run records
struct ScanLayout {
  size_t bits, lkn, lval, lrow, bhead, bcnt, nodes, crow, wtmp, runs, runv, hub, ctrl, total;
};

constexpr uint32_t kHubCap = 64;             // rows of a round whose runs are written by a whole wavefront
// `wave_priv`: wave-private bytes per wavefront; `run_cap`: entries of the plain kernel's run list (0: none); `run_ids`: the
// list also carries every run's row id (self-edge insertion)
__host__ __device__ inline ScanLayout scan_layout(uint32_t bit_words, uint32_t capm, uint32_t nodes_lds, uint32_t waves,
                                                  uint32_t wave_priv = kWavePriv, uint32_t run_cap = 0, bool run_ids = false) {
  ScanLayout L;
  size_t o = 0;
  L.bits = o; o += (size_t)bit_words * 4;
  L.lkn = o; o += r16((size_t)capm * 8);
  L.lval = o; o += r16((size_t)capm * 4);
  L.lrow = o; o += r16((size_t)capm * 4);
  L.bhead = o; o += kSortBuckets * 4;
  L.bcnt = o; o += kSortBuckets * 4;
  L.nodes = o; o += r16((size_t)nodes_lds * 4);
  L.crow = o; o += kMaxRoundChunks * 4;
  L.wtmp = o; o += (size_t)waves * wave_priv;
  L.runs = o; o += (size_t)run_cap * 16;
  L.runv = o; o += run_ids ? (size_t)run_cap * 4 : 0;
  L.hub = o; o += run_cap ? kHubCap * 32 : 0;
  L.ctrl = o; o += C_WORDS * 4;
  L.total = o;
  return L;
}

__device__ __forceinline__ void list_put(const ScanLds &t, uint32_t capm, uint32_t &r, uint32_t key, uint32_t val,
                                         uint32_t row) {
  if (r < capm) { t.lkn[r].x = key; t.lval[r] = val; t.lrow[r] = row; }
  r++;
}

// wave-uniform 64-way search: largest s in [0, P) with a[s] <= x (a non-decreasing, a[0] <= x < a[P])
__device__ __forceinline__ uint32_t find_span(const uint32_t *a, uint32_t P, uint32_t x) {
  uint32_t lo = 0, hi = P;          // answer in [lo, hi)
  const uint32_t lane = lane_id();
  while (hi - lo > 1) {
    const uint32_t span = hi - lo;
    const uint32_t step = (span + 63u) / 64u;
    const uint32_t probe = lo + min(span, (lane + 1u) * step);          // candidates lo+step, lo+2 step, ... (<= hi)
    const bool le = (probe < hi) && (a[probe] <= x);
    const uint64_t m = __ballot(le);
    const uint32_t k = (uint32_t)__popcll(m);                          // monotone: the first k probes are <= x
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
    if (put_over) list_put(t, capm, rr, 2u * w.z + 1u, w.x, r);
  }
}

// membership-filter bit of one id / of the four ids of a quad (bit c = component c); bm4 = (bit_words - 1) * 4
__device__ __forceinline__ uint32_t probe1(const ScanLds &t, uint32_t id, uint32_t bm4) {
  const uint32_t w = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(t.bits) + ((id >> 3) & bm4));
  return __builtin_amdgcn_ubfe(w, id, 1u);
}
__device__ __forceinline__ uint32_t probe4(const ScanLds &t, const uint4 q, uint32_t bm4) {
  // the four reads go out back to back, one wait (left to itself hipcc serialises them: read, wait, read, wait ...)
  const uint32_t base = (uint32_t)(uintptr_t)t.bits;         // LDS byte offset of the filter (low half of the flat address)
  // (the filter sits at LDS offset 0 -- first member of the layout, no static LDS in the scan kernels -- so `| base` = `+ base`
  //  and the address is one v_and_or_b32)
  const uint32_t a0 = ((q.x >> 3) & bm4) | base, a1 = ((q.y >> 3) & bm4) | base;
  const uint32_t a2 = ((q.z >> 3) & bm4) | base, a3 = ((q.w >> 3) & bm4) | base;
  uint32_t w0, w1, w2, w3;
  asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
  return __builtin_amdgcn_ubfe(w0, q.x, 1u) | (__builtin_amdgcn_ubfe(w1, q.y, 1u) << 1) | (__builtin_amdgcn_ubfe(w2, q.z, 1u) << 2) |
         (__builtin_amdgcn_ubfe(w3, q.w, 1u) << 3);
}

constexpr uint32_t kLongRow = 24;            // rows with at least this many quads left in the chunk are streamed alone
constexpr int kRuns = 8;                     // runs (1-KiB loads) a wavefront issues before it consumes the first

// Everything the scan knows about the subgraph / round it works on (uniform), bundled for the group helpers.
struct ScanCtx {
  const uint32_t *indices;
  uint64_t nnz;
  uint32_t *ctrl;
  uint32_t bw_mask, capm;
  bool incl_self, compat;
};

// Probe the four ids of my quad against the membership filter, decide self-edge insertion, append what was found
// to the candidate list.  j0 = (index of my quad's first id) - (row start): wraps when the quad starts in front
// of the row; `edge` (wave-uniform) = 0 promises that every component of every lane is a neighbour of `row`.
template <bool kPlain>
__device__ __forceinline__ void process_group(const ScanCtx &x, const ScanLds &t, const uint4 q, uint32_t j0, uint32_t deg,
                                              uint32_t rs, uint32_t row, uint32_t v, uint32_t e0, uint32_t prev, bool need_prev,
                                              uint32_t edge) {
  const uint32_t cc[4] = {q.x, q.y, q.z, q.w};
  const uint32_t h = probe4(t, q, x.bw_mask << 2);
  uint32_t vmask = 0xFu;
  if (edge) {
    vmask = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (j0 + c < deg) vmask |= 1u << c;
  }
  const uint32_t hit = h & vmask;
  uint32_t selfm = 0, trail = 0;
  if (!kPlain && x.incl_self) {
    // the self edge goes right before the first neighbour > v (.cpp:387-400, :408-410), or after the last one
    const uint32_t up = (uint32_t)__shfl_up((int)cc[3], 1, 64);
    const uint32_t pv0 = need_prev ? prev : up;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t j = j0 + c;
      const uint32_t pv = (c == 0) ? pv0 : cc[c > 0 ? c - 1 : 0];
      const bool prev_lt = (j == 0) || (pv < v);
      if (((vmask >> c) & 1u) && prev_lt && v < cc[c]) selfm |= 1u << c;
      if (((vmask >> c) & 1u) && j + 1u == deg && cc[c] < v) trail = 1;        // last neighbour < v
    }
  }
  // the found ids are few (about one in a hundred): one list reservation per lane, then one entry per set bit
  uint32_t todo = hit | (selfm << 4);
  const uint32_t cnt = (uint32_t)__popc(todo) + trail;
  if (__ballot(cnt != 0)) {
    uint32_t r = 0;
    if (cnt) r = atomicAdd(&x.ctrl[C_M], cnt);
    const uint32_t keybase = 2u * (rs + j0);
    while (todo) {
      const uint32_t b = (uint32_t)__ffs((int)todo) - 1u;
      todo &= todo - 1u;
      const uint32_t c = b & 3u;
      const bool nb = b < 4u;                                                                    // neighbour (.cpp:420-422) / self edge (.cpp:408-410)
      list_put(t, x.capm, r, keybase + 2u * c + (nb ? 1u : 0u), e0 + j0 + c, row);
    }
    if (!kPlain && trail) list_put(t, x.capm, r, 2u * (rs + deg), 0u, row);
  }
  if (!kPlain && x.compat) {
    // reference over-read (.cpp:401-405): when no self edge was inserted INSIDE the row (and none trails),
    // the element right behind the row is examined like a neighbour
    bool fin = false;                          // this lane holds the row's last neighbour
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (((vmask >> c) & 1u) && j0 + c + 1u == deg) fin = true;
    if (fin && !trail) {
      bool inserted = false;
      if (x.incl_self) {
        uint32_t l3 = 0, h3 = deg;
        while (l3 < h3) { const uint32_t m3 = (l3 + h3) >> 1; if (x.indices[e0 + m3] < v) l3 = m3 + 1; else h3 = m3; }
        inserted = !(l3 < deg && x.indices[e0 + l3] == v);
      }
      if (!inserted && (uint64_t)e0 + deg < x.nnz) {
        const uint32_t c = x.indices[e0 + deg];
        if ((t.bits[(c >> 5) & x.bw_mask] >> (c & 31u)) & 1u) {
          uint32_t r = atomicAdd(&x.ctrl[C_M], 1u);
          list_put(t, x.capm, r, 2u * (rs + deg) + 1u, e0 + deg, row);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// end of a round (both scan kernels): resolve the m candidates of the LDS list exactly, order the survivors by key,
// append them to the subgraph's edge scratch, file the round record.  kFromPos: the list holds (position, row)
// only and the key is derived here.
// ---------------------------------------------------------------------------------------------------------------
#define SCAN_T(k) do { if (threadIdx.x == 0) { const uint64_t now_ = clock64(); tacc[k] += (uint32_t)(now_ - tlast); tlast = now_; } } while (0)
// Exact membership of id `cc` in the subgraph's sorted node list: its sub id, or n when absent.  stride == 1: the whole list
// sits in LDS.  stride > 1 (node sets beyond the LDS copy: depth-3 k-hop): LDS holds every stride-th id -- the search runs
// there and ends with ONE global round trip over the <= stride ids of the block it lands in (all loads issued together)
// instead of log2(n) dependent ones.
__device__ __forceinline__ uint32_t node_lookup(const uint32_t *lds_nodes, const uint32_t *g_nodes, uint32_t n, uint32_t stride, uint32_t cc) {
  if (stride == 1u) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (lds_nodes[mid] < cc) lo = mid + 1; else hi = mid; }
    return (lo < n && lds_nodes[lo] == cc) ? lo : n;
  }
  const uint32_t ms = (n + stride - 1u) / stride;
  uint32_t lo = 0, hi = ms;                                // number of samples <= cc
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (lds_nodes[mid] <= cc) lo = mid + 1; else hi = mid; }
  if (lo == 0) return n;
  const uint32_t j0 = (lo - 1u) * stride, j1 = min(n, j0 + stride);
  if (stride <= 16u) {
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = (j0 + (uint32_t)k < j1) ? g_nodes[j0 + k] : kEmpty;
    uint32_t below = 0, hit = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { below += v[k] < cc ? 1u : 0u; hit |= v[k] == cc ? 1u : 0u; }
    return hit ? j0 + below : n;
  }
  uint32_t a = j0, b = j1;
  while (a < b) { const uint32_t mid = (a + b) >> 1; if (g_nodes[mid] < cc) a = mid + 1; else b = mid; }
  return (a < j1 && g_nodes[a] == cc) ? a : n;
}

constexpr uint32_t kSelfEntry = 0x80000000u;     // plain scan: list entry = inserted self edge (row word), lval = position of its slot

template <bool kFromPos>
__device__ __forceinline__ void finish_round(const SampleParams &p, const ScanLds &t, uint32_t *ctrl, uint32_t m, uint32_t n,
                                             uint32_t nstride, const uint32_t *g_nodes, const RowInfo *g_info,
                                             const uint32_t *roots, bool itc, uint32_t *res, uint32_t *g_row, uint32_t *g_col,
                                             uint32_t *g_eid, uint32_t s, uint32_t &rec_blk, uint32_t &rec_cnt, uint32_t *tacc,
                                             uint64_t &tlast) {
  const uint32_t tid = threadIdx.x, T = blockDim.x;
  const uint32_t lane = lane_id(), wave = wave_id();
  const uint32_t cape = p.cap_edges_scr;
  const int R = p.R;
  // ---- resolve the neighbour candidates exactly (.cpp:412-413), order the survivors by key
  // (the keys of a round cover a narrow range: bucket on the offset from the round's smallest key, so that the
  //  256 buckets spread over the round only)
  for (uint32_t i = tid; i < kSortBuckets; i += T) { t.bhead[i] = 0; t.bcnt[i] = 0; }
  if (tid == 0) { ctrl[C_MV] = 0; ctrl[C_CHANGED] = 0xFFFFFFFFu; ctrl[C_MFAIL] = 0; }
  __syncthreads();
  uint32_t kmin_l = kEmpty, kmax_l = 0, alive = 0;
  constexpr int kPer = kFromPos ? 4 : 2;
  for (uint32_t i0 = tid; i0 < m; i0 += (uint32_t)kPer * T) {
    // kPer entries per pass: their global loads (the id at its position, the row record) share one round trip (round 6: four for
    // the flat scan -- a 1 024-thread workgroup then resolves a full 4 096-entry list in ONE pass instead of two)
    uint32_t idx[kPer];
    bool ok[kPer];
    uint32_t key[kPer], pos[kPer], c[kPer];
    uint4 riw[kPer];
#pragma unroll
    for (int e = 0; e < kPer; e++) { idx[e] = i0 + (uint32_t)e * T; ok[e] = idx[e] < m; pos[e] = 0; c[e] = 0; }
#pragma unroll
    for (int e = 0; e < kPer; e++) {
      key[e] = 0; riw[e] = make_uint4(0u, 0u, 0u, 0u);
      if (ok[e]) {
        key[e] = kFromPos ? 1u : t.lkn[idx[e]].x;
        if (kFromPos) {
          // (position, row) entries: neighbours, or -- kSelfEntry in the row word -- the slot of an inserted self edge
          const uint32_t rwf = t.lrow[idx[e]];
          pos[e] = t.lval[idx[e]];
          if (rwf & kSelfEntry) { key[e] = 0u; t.lrow[idx[e]] = rwf & ~kSelfEntry; }
          else c[e] = p.indices[pos[e]];
          riw[e] = *reinterpret_cast<const uint4 *>(g_info + (rwf & ~kSelfEntry));
        } else if (key[e] & 1u) {
          pos[e] = t.lval[idx[e]];
          c[e] = p.indices[pos[e]];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < kPer; e++) {
      if (!ok[e]) continue;
      const uint32_t i = idx[e];
      uint32_t k = key[e];
      if (k & 1u) {
        const uint32_t cc = c[e];
        const uint32_t lo = node_lookup(t.nodes, g_nodes, n, nstride, cc);
        bool keep = lo < n;
        if (keep && !itc && is_root(roots, R, cc)) {
          // multi-root subgraph without include_target_conn: drop root<->root edges (.cpp:414-418)
          const uint32_t rr = t.lrow[i];
          keep = !is_root(roots, R, g_nodes[rr]);
        }
        if (keep && kFromPos) {
          // the scan noted (position, row) only: key = 2 * slot + 1, slot = the row's slot prefix + offset in the row
          k = 2u * (riw[e].z + (pos[e] - riw[e].x)) + 1u;            // RowInfo {e0, deg, rs, v}
          t.lkn[i].x = k;
        }
        if (keep) t.lval[i] = lo;
        else { k = kEmpty; t.lkn[i].x = kEmpty; }
      } else {
        if (kFromPos) {                               // (plain scan: the self edge's key from the position of its slot)
          k = 2u * (riw[e].z + (pos[e] - riw[e].x));
          t.lkn[i].x = k;
        }
        t.lval[i] = t.lrow[i];                        // self edge: column = the row itself
      }
      kmin_l = min(kmin_l, k);                        // (kEmpty is the largest value)
      if (k != kEmpty) { kmax_l = max(kmax_l, k); alive++; }
    }
  }
  {
    // one LDS atomic per wavefront, not per candidate
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      kmin_l = min(kmin_l, (uint32_t)__shfl_xor((int)kmin_l, off, 64));
      kmax_l = max(kmax_l, (uint32_t)__shfl_xor((int)kmax_l, off, 64));
    }
    alive = wave_reduce_sum(alive);
    if (lane == 0 && kmin_l != kEmpty) { atomicMin(&ctrl[C_CHANGED], kmin_l); atomicMax(&ctrl[C_MFAIL], kmax_l); atomicAdd(&ctrl[C_MV], alive); }
  }
  __syncthreads();
  const uint32_t kmin = ctrl[C_CHANGED], kmax = ctrl[C_MFAIL], mv = ctrl[C_MV];
  // the survivors' place in the subgraph's edge scratch: ONE global atomic, its round trip hidden behind the sort
  uint32_t e_base_pending = 0;
  if (tid == 0) e_base_pending = atomicAdd(&res[R_E], mv);
  uint32_t bshift = 0;
  if (kmin != 0xFFFFFFFFu) { while (((kmax - kmin) >> bshift) >= kSortBuckets) bshift++; }
  // counting sort, pass 1: bucket sizes; every survivor keeps its ordinal inside its bucket (upper bits of its row word)
  for (uint32_t i = tid; i < m; i += T) {
    const uint32_t key = t.lkn[i].x;
    if (key != kEmpty) t.lrow[i] |= atomicAdd(&t.bhead[(key - kmin) >> bshift], 1u) << kRankShift;
  }
  __syncthreads();
  SCAN_T(3);
  // exclusive scan of the bucket sizes (kSortBuckets = 256 = 4 per lane of wave 0): bcnt = first place of the bucket
  if (wave == 0) {
    uint32_t b4[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { b4[q] = t.bhead[lane * 4 + q]; sum += b4[q]; }
    const uint32_t incl = wave_incl_scan(sum);
    uint32_t run = incl - sum;
#pragma unroll
    for (int q = 0; q < 4; q++) { t.bcnt[lane * 4 + q] = run; run += b4[q]; }
  }
  __syncthreads();
  // pass 2: the keys, bucket by bucket, into the second word of the list entries (any order inside a bucket)
  for (uint32_t i = tid; i < m; i += T) {
    const uint32_t key = t.lkn[i].x;
    if (key != kEmpty) t.lkn[t.bcnt[(key - kmin) >> bshift] + (t.lrow[i] >> kRankShift)].y = key;
  }
  __syncthreads();
  SCAN_T(9);
  // rank = first place of the bucket + smaller keys inside it: consecutive LDS words, no dependent chain
  for (uint32_t i = tid; i < m; i += T) {
    const uint32_t key = t.lkn[i].x;
    if (key == kEmpty) continue;
    const uint32_t b = (key - kmin) >> bshift;
    const uint32_t base = t.bcnt[b], cnt = t.bhead[b];
    uint32_t r = base;
    uint32_t j = 0;
    // (four independent reads per trip: one LDS wait per four keys instead of one per key -- rank phase 54.7 k -> 46.0 k cycles per
    //  depth-3 segment; eight per trip with a clamped, masked last trip: 51.8 k -- the list reads compete for LDS bandwidth)
    for (; j + 4u <= cnt; j += 4u) {
      const uint32_t a0 = t.lkn[base + j].y, a1 = t.lkn[base + j + 1u].y, a2 = t.lkn[base + j + 2u].y, a3 = t.lkn[base + j + 3u].y;
      r += ((a0 < key) ? 1u : 0u) + ((a1 < key) ? 1u : 0u) + ((a2 < key) ? 1u : 0u) + ((a3 < key) ? 1u : 0u);
    }
    for (; j < cnt; j++) r += (t.lkn[base + j].y < key) ? 1u : 0u;
    t.lrow[i] = (t.lrow[i] & ((1u << kRankShift) - 1u)) | (r << kRankShift);
  }
  SCAN_T(10);
  if (tid == 0) ctrl[C_TICKET] = e_base_pending;
  __syncthreads();
  SCAN_T(11);
  const uint32_t e_base = ctrl[C_TICKET];
  if (tid == 0) {
    // file the round record
    if (rec_cnt == kRecPerBlock) {
      const uint32_t nb = atomicAdd(&p.plan[PL_POOL], 1u);
      if (nb >= p.rec_blocks) { atomicOr(&p.plan[PL_FLAGS], 16u); }
      else { p.blkinfo[rec_blk] = make_uint2(kRecPerBlock, nb); rec_blk = nb; rec_cnt = 0; }
    }
    if (rec_cnt < kRecPerBlock) {
      RoundRec rr;
      rr.s = s; rr.src_off = e_base; rr.cnt = mv; rr.pad = 0;
      p.recs[(size_t)rec_blk * kRecPerBlock + rec_cnt] = rr;
      rec_cnt++;
    }
  }
  for (uint32_t i = tid; i < m; i += T) {
    const uint32_t key = t.lkn[i].x;
    if (key == kEmpty) continue;
    const uint32_t rw_r = t.lrow[i];
    const uint32_t rw = rw_r & ((1u << kRankShift) - 1u);
    const uint32_t o = e_base + (rw_r >> kRankShift);
    if (o < cape) {
      g_row[o] = rw;
      g_col[o] = t.lval[i];
      uint32_t eid = 0xFFFFFFFFu;                                             // inserted self edge (.cpp:410)
      if (key & 1u) { const RowInfo ri = g_info[rw]; eid = ri.e0 + ((key >> 1) - ri.rs); }   // .cpp:422
      g_eid[o] = eid;
    }
  }
}

template <bool kPlain>
__global__ void sg_scan_kernel(SampleParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ScanLayout L = scan_layout(p.bit_words, p.capm, p.nodes_lds, blockDim.x >> 6);
  ScanLds t;
  t.bits = (uint32_t *)(smem + L.bits);
  t.lkn = (uint2 *)(smem + L.lkn);
  t.lval = (uint32_t *)(smem + L.lval);
  t.lrow = (uint32_t *)(smem + L.lrow);
  t.bhead = (uint32_t *)(smem + L.bhead);
  t.bcnt = (uint32_t *)(smem + L.bcnt);
  t.nodes = (uint32_t *)(smem + L.nodes);
  t.crow = (uint32_t *)(smem + L.crow);
  t.wtmp = smem + L.wtmp;
  uint32_t *ctrl = (uint32_t *)(smem + L.ctrl);

  // probe4 forms filter addresses with an OR: the filter must sit on a multiple of its own size in LDS (offset 0 here)
  if (((uint32_t)(uintptr_t)t.bits) & (p.bit_words * 4u - 1u)) __builtin_trap();
  const uint32_t tid = threadIdx.x, T = blockDim.x;
  const uint32_t lane = lane_id(), wave = wave_id(), nw = T >> 6;
  const uint32_t bw_mask = p.bit_words - 1u;
  const uint32_t capm = p.capm, cape = p.cap_edges_scr;
  const int R = p.R;
  const bool incl_self = !kPlain && (p.include_self != 0);
  const bool itc = kPlain || (p.include_target_conn != 0) || (R == 1);   // .cpp:356-358
  const bool compat = !kPlain && (p.compat != 0);
  const bool sentinel = incl_self || compat;
  const uint32_t C = p.plan[PL_NCHUNKS], cpw = p.plan[PL_CPW];
  unsigned char *wflag = t.wtmp + wave * kWavePriv;           // wave-private: 64 start-position flags (all zero between uses) ...
  uint4 *wrun = reinterpret_cast<uint4 *>(wflag + 64);   // ... and the run list of the row window (at most 28 entries, see below)
  for (uint32_t i = tid; i < (T >> 6) * kWavePriv / 4u; i += T) reinterpret_cast<uint32_t *>(t.wtmp)[i] = 0;

  uint32_t tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};               // phase cycles of this workgroup (thread 0)
  uint64_t tlast = clock64();
  // round records of this workgroup (thread 0 files them): block `blockIdx.x`, further blocks from the pool
  uint32_t rec_blk = blockIdx.x, rec_cnt = 0;

  uint32_t g = blockIdx.x * cpw;
  const uint32_t gend = min(C, g + cpw);
  uint32_t s = (g < gend) ? find_span(p.cstart, p.P, g) : 0u;
  while (g < gend) {
    const uint32_t c0 = p.cstart[s], c1 = p.cstart[s + 1];
    if (c1 <= g) { s++; continue; }                       // (subgraphs the selection flagged own no chunk)
    const uint32_t vc0 = g - c0, vc1 = min(c1, gend) - c0;   // this workgroup's (virtual) chunks of subgraph s
    g = c0 + vc1;
    if (vc1 <= p.seg_pad) { s++; continue; }                 // only padding: the next workgroup starts the subgraph
    const uint32_t lc0 = vc0 > p.seg_pad ? vc0 - p.seg_pad : 0u, lc1 = vc1 - p.seg_pad;
    tacc[5]++;
    uint32_t *res = p.s_cnt + (size_t)s * R_WORDS;
    const uint32_t n = res[R_N], Q = res[R_Q];
    const uint32_t *roots = p.roots + (size_t)s * R;
    const uint32_t *g_nodes = p.s_nodes + (size_t)s * p.cap_nodes_scr;
    const RowInfo *g_info = p.s_rowinfo + (size_t)s * p.cap_nodes_scr;
    const uint32_t *g_rowq = p.s_rowq + (size_t)s * (p.cap_nodes_scr + 1);
    uint32_t *g_row = p.s_row + (size_t)s * cape;
    uint32_t *g_col = p.s_col + (size_t)s * cape;
    uint32_t *g_eid = p.s_eid + (size_t)s * cape;
    // (node sets beyond the LDS copy: every nstride-th id is kept -- node_lookup)
    const uint32_t nstride = n <= p.nodes_lds ? 1u : (n + p.nodes_lds - 1u) / p.nodes_lds;
    const uint32_t iq0 = min(Q, lc0 * kQChunk);
    const uint32_t iq1 = min(Q, lc1 * kQChunk);

    // ---- per segment: the subgraph's membership filter (+ its sorted node list, or a sample of it)
    __syncthreads();                                       // (the previous segment's last readers are done)
    for (uint32_t w = tid; w < p.bit_words; w += T) t.bits[w] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += T) {
      const uint32_t v = g_nodes[i];
      atomicOr(&t.bits[(v >> 5) & bw_mask], 1u << (v & 31u));
      if (nstride == 1u) t.nodes[i] = v;
      else if (i % nstride == 0u) t.nodes[i / nstride] = v;
    }
    // (the first barrier of the round loop orders these writes before the scan)

    uint32_t rq0 = iq0;
    uint32_t rquads = min(iq1 - iq0, kMaxRoundChunks * kQChunk);
    for (;;) {
      const uint32_t rq1 = min(iq1, rq0 + max(rquads, 1u));
      const uint32_t rchunks = (rq1 - rq0 + kQChunk - 1u) / kQChunk;
      tacc[6]++;
      SCAN_T(0);
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
      SCAN_T(1);

      // ---- the scan: wave w takes chunks w, w + nw, ... of the round
      uint32_t nx_wb = (wave < rchunks) ? t.crow[wave] : 0u;
      uint32_t nx_base = g_rowq[nx_wb];
      uint4 nx_win = win_fetch(g_info, nx_wb, n);
      for (uint32_t k = wave; k < max(rchunks, 1u); k += nw) {
        const uint32_t qa = rq0 + k * kQChunk, qb = min(qa + kQChunk, rq1);
        // rows without neighbours in front of the subgraph's first quad (or a subgraph without any quad):
        // their sentinel slots belong to the very first chunk of the subgraph
        if (!kPlain && sentinel && qa == 0 && k == 0 && lc0 == 0) {
          const uint32_t stop = (Q == 0) ? n : rl_first(t.crow[0]);
          for (uint32_t r0 = 0; r0 < stop; r0 += 64)
            if (r0 + lane < stop) emit_empty_row<kPlain>(p, t, ctrl, g_info, r0 + lane, incl_self, compat, bw_mask, capm);
        }
        if (rchunks == 0) break;
        // ---- rows are taken 64 at a time (one per lane: the WINDOW), starting at the row that holds the chunk's
        //      first quad.  Every row's quads are clipped to the chunk.  Rows with >= kLongRow quads left are streamed
        //      one row at a time with wave-uniform row scalars -- no position -> row mapping at all, four 1-KiB loads
        //      in flight while a row lasts; the short ones are packed into shared 64-quad groups.
        ScanCtx cx;
        cx.indices = p.indices; cx.nnz = p.nnz; cx.ctrl = ctrl; cx.bw_mask = bw_mask; cx.capm = capm;
        cx.incl_self = incl_self; cx.compat = compat;
        uint32_t wb = rl_first(nx_wb);
        uint32_t wbase = rl_first(nx_base);                            // quad position of the window's first row
        uint4 wcur = nx_win;
        if (k + nw < rchunks) {                                        // the next chunk's first window: loads in flight meanwhile
          nx_wb = t.crow[k + nw];
          nx_base = g_rowq[nx_wb];
          nx_win = win_fetch(g_info, nx_wb, n);
        }
        for (;;) {
          const uint4 wnext = win_fetch(g_info, wb + 64u, n);          // prefetch
          const uint32_t w_e0 = wcur.x, w_deg = wcur.y, w_rs = wcur.z, w_v = wcur.w;
          const uint32_t w_nq = w_deg ? (((w_e0 + w_deg - 1u) >> 2) - (w_e0 >> 2) + 1u) : 0u;
          const uint32_t incl = wave_incl_scan(w_nq);
          const uint32_t W = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
          const uint32_t w_pos = wbase + incl - w_nq;                  // quad position of my row
          const uint32_t lo = max(w_pos, qa), hi = min(w_pos + w_nq, qb);
          const uint32_t w_len = hi > lo ? hi - lo : 0u;               // my row's quads inside the chunk
          const uint32_t w_k0 = lo - w_pos;                            // ... starting at this quad of the row
          // rows without neighbours behind a row that ends inside the chunk: their sentinel slots are this chunk's
          if (!kPlain && sentinel && wb + lane < n && w_nq == 0 && w_pos > qa && w_pos <= qb)
            emit_empty_row<kPlain>(p, t, ctrl, g_info, wb + lane, incl_self, compat, bw_mask, capm);
          // ---- long rows: a flat list of RUNS (<= 64 quads of one row).  The rows write their runs' records to a
          //      wave-private LDS list in parallel (sum of ceil(len / 64) over rows of >= kLongRow quads in 512: at most 28); it then issues
          //      up to kRuns 1-KiB loads back to back -- across rows -- before it consumes the first: with 16 wavefronts
          //      per CU that keeps the ~64 KiB in flight this access pattern needs (scripts/probe_row_stream.py).
          {
            const bool is_long = w_len >= kLongRow;
            const uint32_t nr = is_long ? (w_len + 63u) >> 6 : 0u;
            const uint32_t rincl = wave_incl_scan(nr);
            const uint32_t nruns = (uint32_t)__builtin_amdgcn_readlane((int)rincl, 63);
            if (nruns) {
              uint32_t ri = rincl - nr;
              for (uint32_t g0 = 0; g0 < (is_long ? w_len : 0u); g0 += 64u, ri++) {
                const uint32_t take = min(64u, w_len - g0);
                const uint32_t kq = w_k0 + g0;
                // every component is a neighbour unless the run holds the row's first / last quad or is not full
                const uint32_t edge = ((kq == 0 && (w_e0 & 3u)) || (kq + take == w_nq && ((w_e0 + w_deg) & 3u)) || take < 64u) ? 1u : 0u;
                // record: first id of the run, (row lane | quads - 1 | edge | not the row's first quad), key of id 0
                wrun[ri] = make_uint4(((w_e0 >> 2) + kq) << 2, lane | ((take - 1u) << 6) | (edge << 12) | (kq > 0 ? 1u << 13 : 0u),
                                      2u * (w_rs - w_e0) + 1u, 0u);
              }
              __builtin_amdgcn_wave_barrier();
              for (uint32_t i0 = 0; i0 < nruns; i0 += kRuns) {
                const uint32_t nfill = min((uint32_t)kRuns, nruns - i0);
                const uint4 dl = wrun[i0 + (lane & (kRuns - 1))];          // lane u holds run i0 + u
                uint4 q[kRuns];
                uint32_t pv[kRuns];
#pragma unroll
                for (int u = 0; u < kRuns; u++) {
                  q[u] = make_uint4(kEmpty, kEmpty, kEmpty, kEmpty);
                  pv[u] = 0;
                  if ((uint32_t)u < nfill) {                                // wave-uniform
                    const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)dl.x, u);
                    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)dl.y, u);
                    if (lane <= ((m >> 6) & 63u)) q[u] = *reinterpret_cast<const uint4 *>(p.indices + a0 + 4u * lane);
                    if (!kPlain && incl_self && lane == 0 && (m & (1u << 13))) pv[u] = p.indices[a0 - 1u];
                  }
                }
                if (kPlain) {
                  // probe all runs first (4 bits per run and lane), then file what was found -- about one id in a
                  // hundred -- in one go: one list reservation per batch, the ids noted by position
                  uint32_t hm = 0;
#pragma unroll
                  for (int u = 0; u < kRuns; u++) {
                    if ((uint32_t)u < nfill) {
                      const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)dl.y, u);
                      uint32_t h = probe4(t, q[u], bw_mask << 2);
                      if (m & (1u << 12)) {                                  // the run holds an end of its row / is not full
                        const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)dl.x, u);
                        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)w_e0, m & 63u);
                        const uint32_t deg = (uint32_t)__builtin_amdgcn_readlane((int)w_deg, m & 63u);
                        const uint32_t j0 = a0 + 4u * lane - e0;            // wraps in front of the row
                        const uint32_t dq = lane <= ((m >> 6) & 63u) ? deg : 0u;
                        uint32_t vmask = 0;
#pragma unroll
                        for (int c = 0; c < 4; c++)
                          if (j0 + c < dq) vmask |= 1u << c;
                        h &= vmask;
                      }
                      hm |= h << (4 * u);
                    }
                  }
                  const uint32_t cnt = (uint32_t)__popc(hm);
                  if (__ballot(cnt != 0)) {
                    const uint32_t incl = wave_incl_scan(cnt);
                    uint32_t base = 0;
                    if (lane == 63) base = atomicAdd(&ctrl[C_M], incl);
                    uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)base, 63) + incl - cnt;
                    while (hm) {
                      const uint32_t b = (uint32_t)__ffs((int)hm) - 1u;
                      hm &= hm - 1u;
                      const uint4 d = wrun[i0 + (b >> 2)];
                      const uint32_t idx = d.x + 4u * lane + (b & 3u);
                      list_put(t, capm, r, d.z + 2u * idx, idx, wb + (d.y & 63u));        // .cpp:420-422
                    }
                  }
                } else {
#pragma unroll
                  for (int u = 0; u < kRuns; u++) {
                    if ((uint32_t)u < nfill) {
                      const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)dl.x, u);
                      const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)dl.y, u);
                      const uint32_t rl = m & 63u;
                      const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)w_e0, rl);
                      const uint32_t deg = (uint32_t)__builtin_amdgcn_readlane((int)w_deg, rl);
                      const uint32_t rs = (uint32_t)__builtin_amdgcn_readlane((int)w_rs, rl);
                      const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)w_v, rl);
                      process_group<kPlain>(cx, t, q[u], a0 + 4u * lane - e0, lane <= ((m >> 6) & 63u) ? deg : 0u, rs, wb + rl, v, e0,
                                            pv[u], lane == 0 && (m & (1u << 13)), (m >> 12) & 1u);
                    }
                  }
                }
              }
              __builtin_amdgcn_wave_barrier();
            }
          }
          // ---- short rows, packed: the rows flag the packed position they start on, the positions read the flags
          //      back and fill forward (DPP max-scan), lane gathers fetch the row's scalars
          {
            const bool shortrow = w_len != 0 && w_len < kLongRow;
            const uint32_t tl = shortrow ? w_len : 0u;
            const uint32_t tincl = wave_incl_scan(tl);
            const uint32_t tq = tincl - tl;
            const uint32_t ttot = (uint32_t)__builtin_amdgcn_readlane((int)tincl, 63);
            for (uint32_t p0 = 0; p0 < ttot; p0 += 64u) {
              const uint32_t take = min(64u, ttot - p0);
              const bool starts = shortrow && tq > p0 && tq < p0 + take;
              if (starts) wflag[tq - p0] = (unsigned char)(lane + 1u);
              __builtin_amdgcn_wave_barrier();
              uint32_t rl = wflag[lane];
              __builtin_amdgcn_wave_barrier();
              if (starts) wflag[tq - p0] = 0;
              const uint64_t before = __ballot(shortrow && tq <= p0);
              const uint32_t f0 = 63u - (uint32_t)__builtin_clzll((unsigned long long)(before | 1ull));   // row that holds p0
              if (lane == 0) rl = f0 + 1u;
              rl = wave_incl_max_scan(rl) - 1u;                              // lane of the window that owns my position
              const uint32_t e0 = lane_get(w_e0, rl), rtq = lane_get(tq, rl), rk0 = lane_get(w_k0, rl);
              const uint32_t deg = lane_get(w_deg, rl), rs = lane_get(w_rs, rl);
              const uint32_t v = kPlain ? 0u : lane_get(w_v, rl);
              const uint32_t kq = rk0 + (p0 + lane - rtq);                   // quad index inside the row
              const uint32_t a = ((e0 >> 2) + kq) << 2;
              uint4 q = make_uint4(kEmpty, kEmpty, kEmpty, kEmpty);
              uint32_t pv = 0;
              if (lane < take) q = *reinterpret_cast<const uint4 *>(p.indices + a);
              // lane - 1 holds the previous quad of my row unless my quad opens the row's run in this group
              const bool np = lane < take && kq > 0 && (lane == 0 || p0 + lane == rtq);
              if (!kPlain && incl_self && np) pv = p.indices[a - 1u];
              process_group<kPlain>(cx, t, q, a - e0, lane < take ? deg : 0u, rs, wb + rl, v, e0, pv, np, 1u);
            }
          }
          // ---- next window?  Done when a row with neighbours starts at or behind the chunk's end (rows without
          //      neighbours sitting exactly on the chunk's end still belong to it, wherever their window is)
          const uint64_t stop = __ballot(wb + lane < n && w_nq != 0 && w_pos >= qb);
          if (stop || wb + 64u >= n || wbase + W > qb) break;
          wb += 64u;
          wbase += W;
          wcur = wnext;
        }
      }
      __syncthreads();
      SCAN_T(2);
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
      finish_round<false>(p, t, ctrl, m, n, nstride, g_nodes, g_info, roots, itc, res, g_row, g_col, g_eid, s, rec_blk, rec_cnt,
                          tacc, tlast);
      rq0 = rq1;
      __syncthreads();
      SCAN_T(4);
      if (rq0 >= iq1) break;
    }
    s++;
  }
  if (tid == 0) {
    p.blkinfo[rec_blk] = make_uint2(rec_cnt, 0xFFFFFFFFu);
    tacc[8] = tacc[0] + tacc[1] + tacc[2] + tacc[3] + tacc[4];          // (word 8: the LONGEST workgroup's phase cycles -- balance of the span partition)
    for (int i = 0; i < 12; i++) {
      if (i == 8) atomicMax(&p.plan[PL_T0 + i], tacc[i] >> 4);
      else atomicAdd(&p.plan[PL_T0 + i], (i < 5 || i > 6) ? tacc[i] >> 4 : tacc[i]);   // cycles in units of 16
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The plain scan (no self-edge insertion, no compat over-read, single root or include_target_conn): the benchmark's
// case, built around ONE flat run list per round so that the streaming loop has no row logic in it.
//
//   A. one thread per row of the subgraph clips its quads to the round and counts its RUNS (<= 64 quads of one row); a
//      block scan places them; every row writes its 16-byte run records {first id position, row | quads | edge flag,
//      row begin, row end} into the LDS list (rows with many runs hand them to a whole wavefront).
//   B. wavefront w streams runs w, w + nw, ...: eight 1-KiB loads are always in flight -- the loop is software
//      pipelined in straight-line code (the load of run j + 8 is issued right after run j was probed; loads are never
//      predicated or branched around, so the counted `s_waitcnt vmcnt(7)` the compiler derives is exact; scheduling
//      barriers keep the slots in issue order).  Lanes beyond a run's length re-read its last quad and are masked, so a
//      short row costs one instruction slot but only its own cache lines.  Found ids -- about one in a hundred -- are
//      filed once per eight runs: one DPP scan + one LDS atomic per wavefront, entries (position, row).
//   C. finish_round<true>: ids fetched by position, exact membership, keys derived, bucket sort, write-out.
//
// Measured against the row-window kernel above on the same box (scripts/ab_scan.sh, products shape): 0.193 vs 0.257 ms
// at 1 024 roots, 0.995 vs 1.135 ms at 8 192; arxiv shape (rows of ~3 quads) 0.100 vs 0.143 ms at 256 roots.
// ---------------------------------------------------------------------------------------------------------------
constexpr int C_NRUN = C_NNODES, C_NHUB = C_NF0, C_ROWSTOP = 12 /* and 13: one word per pass parity */, C_ROWNEXT = 14;
constexpr uint32_t kHubRuns = 8;             // rows with at least this many runs in the round are expanded by a wavefront
constexpr int kDepth = 8;                    // 1-KiB loads a wavefront keeps in flight

__device__ __forceinline__ uint4 make_run(const RowInfo ri, uint32_t row, uint32_t nq, uint32_t k0, uint32_t len, uint32_t g0) {
  const uint32_t take = min(64u, len - g0);
  const uint32_t kq = k0 + g0;
  // every component is a neighbour unless the run holds the row's first / last quad
  const uint32_t edge = ((kq == 0 && (ri.e0 & 3u)) || (kq + take == nq && ((ri.e0 + ri.deg) & 3u))) ? 1u : 0u;
  return make_uint4(((ri.e0 >> 2) + kq) << 2, row | ((take - 1u) << 20) | (edge << 26), ri.e0, ri.e0 + ri.deg);
}

// slot[v] = the position in `indices` where the reference inserts v's self edge: in front of the first neighbour above v, behind the
// last one when all lie below (== e1), kEmpty when the row lists v itself (lower_bound != upper_bound, ParallelSampler.cpp:386-400)
__global__ void sg_self_slot_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices, uint32_t N,
                                    uint32_t *__restrict__ slot) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  uint32_t lo = indptr[v];
  const uint32_t e1 = indptr[v + 1];
  uint32_t hi = e1;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (indices[mid] < v) lo = mid + 1u; else hi = mid;
  }
  slot[v] = (lo < e1 && indices[lo] == v) ? kEmpty : lo;
}

template <bool kSelf>
__global__ void sg_scan_plain_kernel(SampleParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ScanLayout L = scan_layout(p.bit_words, p.capm, p.nodes_lds, blockDim.x >> 6, 0u, p.run_cap, kSelf);
  ScanLds t;
  t.bits = (uint32_t *)(smem + L.bits);
  t.lkn = (uint2 *)(smem + L.lkn);
  t.lval = (uint32_t *)(smem + L.lval);
  t.lrow = (uint32_t *)(smem + L.lrow);
  t.bhead = (uint32_t *)(smem + L.bhead);
  t.bcnt = (uint32_t *)(smem + L.bcnt);
  t.nodes = (uint32_t *)(smem + L.nodes);
  t.crow = (uint32_t *)(smem + L.crow);
  t.wtmp = smem + L.wtmp;
  uint4 *runs = (uint4 *)(smem + L.runs);
  uint4 *hub = (uint4 *)(smem + L.hub);
  uint32_t *ctrl = (uint32_t *)(smem + L.ctrl);

  // probe4 forms filter addresses with an OR: the filter must sit on a multiple of its own size in LDS (offset 0 here)
  if (((uint32_t)(uintptr_t)t.bits) & (p.bit_words * 4u - 1u)) __builtin_trap();
  const uint32_t tid = threadIdx.x, T = blockDim.x;
  const uint32_t lane = lane_id(), wave = wave_id(), nw = T >> 6;
  const uint32_t bw_mask = p.bit_words - 1u, bm4 = bw_mask << 2;
  const uint32_t capm = p.capm, cape = p.cap_edges_scr, run_cap = p.run_cap;
  const int R = p.R;
  const uint32_t C = p.plan[PL_NCHUNKS], cpw = p.plan[PL_CPW];

  uint32_t tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tlast = clock64();
  uint32_t rec_blk = blockIdx.x, rec_cnt = 0;

  uint32_t g = blockIdx.x * cpw;
  const uint32_t gend = min(C, g + cpw);
  uint32_t s = (g < gend) ? find_span(p.cstart, p.P, g) : 0u;
  while (g < gend) {
    const uint32_t c0 = p.cstart[s], c1 = p.cstart[s + 1];
    if (c1 <= g) { s++; continue; }
    const uint32_t vc0 = g - c0, vc1 = min(c1, gend) - c0;   // this workgroup's (virtual) chunks of subgraph s
    g = c0 + vc1;
    if (vc1 <= p.seg_pad) { s++; continue; }                 // only padding: the next workgroup starts the subgraph
    const uint32_t lc0 = vc0 > p.seg_pad ? vc0 - p.seg_pad : 0u, lc1 = vc1 - p.seg_pad;
    tacc[5]++;
    uint32_t *res = p.s_cnt + (size_t)s * R_WORDS;
    const uint32_t n = res[R_N], Q = res[R_Q];
    const uint32_t *roots = p.roots + (size_t)s * R;
    const uint32_t *g_nodes = p.s_nodes + (size_t)s * p.cap_nodes_scr;
    const RowInfo *g_info = p.s_rowinfo + (size_t)s * p.cap_nodes_scr;
    const uint32_t *g_rowq = p.s_rowq + (size_t)s * (p.cap_nodes_scr + 1);
    const uint32_t *g_selfpos = kSelf ? p.s_selfpos + (size_t)s * p.cap_nodes_scr : nullptr;
    uint32_t *g_row = p.s_row + (size_t)s * cape;
    uint32_t *g_col = p.s_col + (size_t)s * cape;
    uint32_t *g_eid = p.s_eid + (size_t)s * cape;
    const uint32_t nstride = n <= p.nodes_lds ? 1u : (n + p.nodes_lds - 1u) / p.nodes_lds;
    const uint32_t iq0 = min(Q, lc0 * kQChunk);
    const uint32_t iq1 = min(Q, lc1 * kQChunk);

    // ---- per segment: the subgraph's membership filter (+ its sorted node list, or a sample of it)
    __syncthreads();
    for (uint32_t w = tid; w < p.bit_words; w += T) t.bits[w] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += T) {
      const uint32_t v = g_nodes[i];
      atomicOr(&t.bits[(v >> 5) & bw_mask], 1u << (v & 31u));
      if (nstride == 1u) t.nodes[i] = v;
      else if (i % nstride == 0u) t.nodes[i / nstride] = v;
    }
    // The rows are ordered by quad position: a round only looks at the rows from the one that holds its first quad on, in
    // passes of T rows, until a row starts behind its end (a depth-3 subgraph has 4 600 rows and ~20 rounds per segment:
    // walking all rows every round was half the kernel's time).  row_lo: found by search once per segment, then carried
    // from round to round.
    uint32_t row_lo = (iq0 == 0u || iq0 >= Q) ? (iq0 == 0u ? 0u : n) : find_span(g_rowq, n, iq0);

    // runs of a round <= quads / 64 + rows with >= kLongRow quads: start with a round that fits for sure
    uint32_t rq0 = iq0;
    uint32_t rquads = min(kMaxRoundChunks, max(1u, (run_cap - min(n, run_cap / 2u)) / (kQChunk / 64u))) * kQChunk;
    for (;;) {
      const uint32_t rq1 = min(iq1, rq0 + rquads);
      tacc[6]++;
      SCAN_T(0);
      if (tid == 0) { ctrl[C_M] = 0; ctrl[C_NHUB] = 0; ctrl[C_ROWSTOP] = 0; ctrl[C_ROWSTOP + 1] = 0; ctrl[C_ROWNEXT] = row_lo; }
      __syncthreads();
      // ---- A. the round's run list (+ the inserted self edges whose slot lies in the round, ParallelSampler.cpp:386-400)
      uint32_t nrun = 0;
      uint32_t pass = 0;
      for (uint32_t base = row_lo; base < n; base += T, pass ^= 1u) {
        const uint32_t r = base + tid;
        uint32_t nr = 0, len = 0, k0 = 0, nq = 0, qs = 0xFFFFFFFFu;
        RowInfo ri;
        ri.e0 = ri.deg = ri.rs = ri.v = 0;
        if (r < n) {
          // (kSelf: the row's self-edge slot comes from the array the selection kernel filled beside the row records -- the same
          //  coalesced round trip as the record; the per-node table lookup here was a dependent random gather per round)
          uint32_t sp = kEmpty;
          if (kSelf) sp = g_selfpos[r];
          qs = g_rowq[r];
          const uint32_t qe = g_rowq[r + 1];
          const uint4 riw = *reinterpret_cast<const uint4 *>(g_info + r);
          ri.e0 = riw.x; ri.deg = riw.y; ri.rs = riw.z; ri.v = riw.w;
          const uint32_t lo = max(qs, rq0), hi = min(qe, rq1);
          nq = qe - qs;
          if (hi > lo) { len = hi - lo; k0 = lo - qs; nr = (len + 63u) >> 6; }
          if (kSelf) {
            // The inserted self edge (.cpp:386-400) sits in front of the first neighbour above the row's own id v -- behind the last
            // one when all lie below -- and a row that lists itself gets none: a property of the FULL graph's row, looked up in the
            // per-node table the handle built once (SampleParams::self_slot; rounds 5 found the place where the ids stream by:
            // ~200 scalar instructions per run, a third of the depth-3 scan).  The entry is filed by the round that scans the quad
            // its position lies in (the last quad for a slot behind the row), so that it is ranked with that quad's survivors.
            if (sp != kEmpty) {
              bool mine;
              if (nq == 0u) {
                // a row WITHOUT neighbours: the round whose range ends at or behind its (empty) position -- the very first round
                // of the subgraph for position 0, as sg_scan_kernel files them
                mine = (qs > rq0 && qs <= rq1) || (qs == 0u && rq0 == 0u && lc0 == 0u);
              } else {
                const uint32_t pos = sp >= ri.e0 + ri.deg ? ri.e0 + ri.deg - 1u : sp;
                const uint32_t qi = qs + ((pos >> 2) - (ri.e0 >> 2));
                mine = qi >= rq0 && qi < rq1;
              }
              if (mine) {
                const uint32_t idx = atomicAdd(&ctrl[C_M], 1u);
                if (idx < capm) { t.lval[idx] = sp; t.lrow[idx] = r | kSelfEntry; }
              }
            }
          }
        }
        // (rows ascend: the last lane of a wavefront whose row starts in front of the round's end is the wavefront's largest)
        {
          const uint64_t before = __ballot(r < n && qs < rq1);
          if (before && lane == 63u - (uint32_t)__builtin_clzll((unsigned long long)before)) atomicMax(&ctrl[C_ROWNEXT], r);
          // (one stop word per pass parity: a wavefront that is already in the next pass must not change what a slower one
          //  still has to read -- the scan's barriers keep everybody within one pass of each other)
          if (__ballot(r >= n || qs > rq1) && lane == 0) ctrl[C_ROWSTOP + pass] = 1u;
        }
        uint32_t tot;
        const uint32_t first = nrun + block_excl_scan(nr, t.bhead, &tot);
        nrun += tot;
        if (nr && first + nr <= run_cap) {
          uint32_t h = kHubCap;
          if (nr >= kHubRuns) h = atomicAdd(&ctrl[C_NHUB], 1u);
          if (h < kHubCap) { hub[2 * h] = make_uint4(r, first, k0, len); hub[2 * h + 1] = make_uint4(ri.e0, ri.deg, nq, ri.v); }
          else {
            for (uint32_t g0 = 0, i = first; g0 < len; g0 += 64u, i++) runs[i] = make_run(ri, r, nq, k0, len, g0);
          }
        }
        if (ctrl[C_ROWSTOP + pass]) break;                // (written before the scan's barriers, uniform behind them)
      }
      __syncthreads();
      const uint32_t row_next = ctrl[C_ROWNEXT];
      if (nrun > run_cap) {
        // too many rows start in the round: halve it (a run holds at least one quad, so 64 quads always fit: the host
        // keeps run_cap >= 128)
        rquads = max((rq1 - rq0) / 2u, 64u);
        __syncthreads();
        continue;
      }
      {
        const uint32_t nhub = min(ctrl[C_NHUB], kHubCap);
        for (uint32_t h = wave; h < nhub; h += nw) {
          const uint4 hb = hub[2 * h], hr = hub[2 * h + 1];
          RowInfo ri;
          ri.e0 = hr.x; ri.deg = hr.y; ri.rs = 0; ri.v = 0;
          const uint32_t nq = hr.z;
          for (uint32_t j = lane; j * 64u < hb.w; j += 64u) runs[hb.y + j] = make_run(ri, hb.x, nq, hb.z, hb.w, j * 64u);
        }
      }
      __syncthreads();
      SCAN_T(1);

      // ---- B. stream the runs: wavefront w takes runs w, w + nw, ...
      if (wave < nrun) {
        const uint32_t J = (nrun - wave + nw - 1u) / nw;
        uint4 q[kDepth];
        uint32_t r_meta[kDepth], r_e0[kDepth], r_e1[kDepth], r_a0[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; u++) {
          const uint32_t i = wave + (uint32_t)u * nw;
          const uint4 rc = runs[min(i, nrun - 1u)];
          r_a0[u] = rl_first(rc.x); r_e0[u] = rl_first(rc.z); r_e1[u] = rl_first(rc.w);
          r_meta[u] = rl_first(i < nrun ? rc.y : 0xFFFFFFFFu);                        // (all ones: no such run)
          const uint32_t tk = r_meta[u] == 0xFFFFFFFFu ? 0u : (r_meta[u] >> 20) & 63u;
          q[u] = *reinterpret_cast<const uint4 *>(p.indices + r_a0[u] + 4u * min(lane, tk));
          __builtin_amdgcn_sched_barrier(0);                  // (keep the slots in issue order: the counted waits below rely on it)
        }
        for (uint32_t j0 = 0; j0 < J; j0 += kDepth) {
          uint32_t hm = 0;
#pragma unroll
          for (int u = 0; u < kDepth; u++) {
            // consume run j0 + u ...
            const uint32_t meta = r_meta[u];
            uint32_t h = probe4(t, q[u], bm4);
            const uint32_t tk = (meta >> 20) & 63u;
            if (meta == 0xFFFFFFFFu) h = 0;
            else if (tk != 63u || (meta & (1u << 26))) {
              // an end of the row inside the run, or lanes beyond its length: components [lo, hi) of my quad are the row's
              const uint32_t i0 = r_a0[u] + 4u * min(lane, tk);
              const int lo = min(max((int)(r_e0[u] - i0), 0), 4), hi = min(max((int)(r_e1[u] - i0), 0), 4);
              uint32_t vm = (0xFu << lo) & ~(0xFu << hi) & 0xFu;
              if (lane > tk) vm = 0;
              h &= vm;
            }
            hm |= h << (4 * u);
            // ... and put run j0 + u + kDepth in its place
            const uint32_t i2 = wave + (j0 + (uint32_t)u + kDepth) * nw;
            const uint4 rc = runs[min(i2, nrun - 1u)];
            r_a0[u] = rl_first(rc.x); r_e0[u] = rl_first(rc.z); r_e1[u] = rl_first(rc.w);
            r_meta[u] = rl_first(i2 < nrun ? rc.y : 0xFFFFFFFFu);
            const uint32_t tk2 = r_meta[u] == 0xFFFFFFFFu ? 0u : (r_meta[u] >> 20) & 63u;
            q[u] = *reinterpret_cast<const uint4 *>(p.indices + r_a0[u] + 4u * min(lane, tk2));
            __builtin_amdgcn_sched_barrier(0);
          }
          const uint32_t cnt = (uint32_t)__popc(hm);
          if (__ballot(cnt != 0)) {
            const uint32_t incl = wave_incl_scan(cnt);
            uint32_t basev = 0;
            if (lane == 63) basev = atomicAdd(&ctrl[C_M], incl);
            uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)basev, 63) + incl - cnt;
            while (hm) {
              const uint32_t b = (uint32_t)__ffs((int)hm) - 1u;
              hm &= hm - 1u;
              const uint4 d = runs[wave + (j0 + (b >> 2)) * nw];
              if (r < capm) { t.lval[r] = d.x + 4u * lane + (b & 3u); t.lrow[r] = d.y & ((1u << kRankShift) - 1u); }   // .cpp:420-422
              r++;
            }
          }
        }
      }
      SCAN_T(7);
      __syncthreads();
      SCAN_T(2);
      const uint32_t m = ctrl[C_M];
      if (m > capm) {
        // too many candidates for the list: redo this round on half the quads
        if (rq1 - rq0 <= 64u) {
          if (tid == 0) atomicOr(&res[R_FLAGS], 2u);
          break;
        }
        rquads = max((rq1 - rq0) / 2u, 64u);
        __syncthreads();
        continue;
      }
      finish_round<true>(p, t, ctrl, m, n, nstride, g_nodes, g_info, roots, true, res, g_row, g_col, g_eid, s, rec_blk, rec_cnt,
                         tacc, tlast);
      rq0 = rq1;
      row_lo = row_next;                                // (the last row that starts in front of the round's end may reach into the next)
      __syncthreads();
      SCAN_T(4);
      if (rq0 >= iq1) break;
    }
    s++;
  }
  if (tid == 0) {
    p.blkinfo[rec_blk] = make_uint2(rec_cnt, 0xFFFFFFFFu);
    tacc[8] = tacc[0] + tacc[1] + tacc[2] + tacc[3] + tacc[4];          // (word 8: the LONGEST workgroup's phase cycles -- balance of the span partition)
    for (int i = 0; i < 12; i++) {
      if (i == 8) atomicMax(&p.plan[PL_T0 + i], tacc[i] >> 4);
      else atomicAdd(&p.plan[PL_T0 + i], (i < 5 || i > 6) ? tacc[i] >> 4 : tacc[i]);   // cycles in units of 16
    }
  }
}

#undef SCAN_T

}  // namespace shadow
