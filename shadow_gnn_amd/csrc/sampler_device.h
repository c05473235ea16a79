// sampler_device.h -- device side of the MI355X subgraph sampler, part 1: node selection.
//
// The sampler is a short pipeline of kernels per call (host side: sampler.hip):
//
//   sg_select_lds_kernel   one WORKGROUP per subgraph (persistent, ticket): phases 1-3 below, tables in LDS
//   sg_select_big_kernel   the same over global-memory tables for node sets beyond the LDS tables
//   sg_plan_kernel         cuts every subgraph's quad stream into work items (sampler_scan.h)
//   sg_scan_kernel         the node-induced slicing as a FLAT grid over (subgraph, quad range) items,
//                          every CU streaming neighbour ids all the time (sampler_scan.h)
//   sg_relocate_kernel     block-diagonal assembly, row pointers, hop BFS / DRNL (sampler.hip)
//
// Phases of the selection kernel:
//   1. node selection: k-hop frontier expansion (ParallelSampler.cpp:510-547),
//      PPR top-k (.cpp:565-590) or the roots alone (.cpp:498-505).  Dedupe goes
//      through an LDS hash set of 16-byte buckets (4 keys per bucket: a lookup is
//      ONE ds_read_b128 and four compares, no probe loop) with a tiny overflow
//      stash; frontiers are LDS lists; budgeted draws are Philox4x32-10.
//   2. ids sorted ascending in LDS (== std::sort, .cpp:362); the position in the sorted
//      list is the sub id (orig2subID, .cpp:369-372).
//   3. per row: slot prefix (deg+1 slots: position keys), quad prefix (aligned
//      16-byte quads of the row in the indices array) and row start -- written to the
//      subgraph's scratch as one 16-byte record per row for the scan kernel.
#pragma once
#include "common.h"

namespace shadow {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr uint32_t kQChunk = 512;        // quads a wavefront scans per round (= 2048 neighbour ids, 8 x 1 KiB loads)
constexpr uint32_t kLdsCapNodes = 2048;  // largest node set handled by the LDS selection kernel / kept in LDS by the scan
constexpr uint32_t kMaxRoots = 8;
constexpr uint32_t kStash = 64;          // overflow entries behind the bucket array
constexpr uint32_t kSortBuckets = 256;
constexpr uint32_t kBitWords = 8192;     // scan membership filter: 256 Ki bits indexed by the low id bits (32 KB of LDS)
constexpr uint32_t kBitWordsBig = 32768; // same for node sets beyond the LDS tables (10^3..10^5 nodes): 1 Mi bits

// control words (LDS)
enum { C_NNODES = 0, C_NF0 = 1, C_NF1 = 2, C_OVF = 3, C_MFAIL = 4, C_FRONT_NODES = 5,
       C_FRONT_READS = 6, C_CHANGED = 7, C_M = 8, C_TICKET = 9, C_NSTASH = 10, C_MV = 11, C_WORDS = 16 };

// per-subgraph result words in scratch (s_cnt)
// (R_E doubles as the bump allocator of the subgraph's edge scratch while the scan kernel runs)
enum { R_N = 0, R_E = 1, R_FLAGS = 2, R_SLOTS = 3, R_FNODES = 4, R_FREADS = 5, R_Q = 6, R_T0 = 8, R_WORDS = 16 };

// One 16-byte record per row (= node, in ascending id order) of a subgraph, written by the selection kernel,
// read by the scan kernel with wave-uniform loads.
struct RowInfo {
  uint32_t e0;    // start of the node's row in the full graph's indices array
  uint32_t deg;   // full-graph degree
  uint32_t rs;    // slot prefix: sum over the preceding rows of (deg + 1); slot rs + j = neighbour j, rs + deg = sentinel
  uint32_t v;     // original node id
};

// Round record of the scan kernel: the survivors of one round of subgraph `s`, in key order, live at
// [src_off, src_off + cnt) of that subgraph's edge scratch.  Records are filed in blocks of kRecPerBlock: scan
// workgroup w owns block w and chains further blocks from a pool (blkinfo[b] = {records in block b, next block}).
struct RoundRec {
  uint32_t s, src_off, cnt, pad;
};

// plan words (global): written by sg_plan_kernel
// (PL_T0..: cycles workgroup leaders spent per scan phase -- setup, start rows, scan, resolve, sort + write, items, rounds)
enum { PL_NCHUNKS = 0, PL_CPW = 1, PL_POOL = 2, PL_FLAGS = 3, PL_T0 = 8, PL_WORDS = 24 };

struct SampleParams {
  const uint32_t *indptr;
  const uint32_t *indices;
  uint32_t N;
  uint64_t nnz;
  const uint32_t *roots;  // [P*R]
  uint32_t P;
  int R;
  int method, depth, budget, k;
  float threshold;
  int include_self, include_target_conn, compat;
  uint64_t seed, serial_base;
  const int32_t *ppr_row;
  const uint32_t *ppr_len;
  const uint32_t *ppr_neigh;
  const float *ppr_score;
  uint32_t ppr_stride;
  // table geometry of the selection kernel
  uint32_t capn;      // node capacity of the tables used by this launch
  uint32_t capf;      // frontier list capacity
  uint32_t H;         // hash key slots in buckets (multiple of 4, power of two)
  uint32_t hshift;    // 32 - log2(H/4)
  // scan kernel geometry
  uint32_t capm;      // capacity of the LDS candidate list
  uint32_t bit_words; // membership filter size in words (power of two)
  uint32_t nodes_lds; // node ids of subgraphs up to this size are kept in LDS for the candidate resolution
  uint32_t run_cap;   // entries of the plain scan kernel's LDS run list
  uint32_t seg_pad;   // chunks' worth of work a subgraph segment costs a scan workgroup before it streams anything
  // per-subgraph scratch (stride = cap_nodes_scr / cap_edges_scr)
  uint32_t cap_nodes_scr, cap_edges_scr;
  uint32_t *s_nodes;   // [P*cap_nodes_scr] sorted node ids
  float *s_ppr;        // [P*cap_nodes_scr]
  RowInfo *s_rowinfo;  // [P*cap_nodes_scr]
  uint32_t *s_rowq;    // [P*(cap_nodes_scr+1)] quad prefix
  uint32_t *s_row;     // [P*cap_edges_scr] local row of each emitted edge
  uint32_t *s_col;     // [P*cap_edges_scr]
  uint32_t *s_eid;     // [P*cap_edges_scr]
  uint32_t *s_tgt;     // [P*kMaxRoots]
  uint32_t *s_cnt;     // [P*R_WORDS]
  // work partition of the scan
  uint32_t *cstart;    // [P+1] first chunk of every subgraph in the global chunk sequence
  uint32_t *plan;      // [PL_WORDS]
  RoundRec *recs;      // [rec_blocks * kRecPerBlock] round records
  uint2 *blkinfo;      // [rec_blocks] {records filed in the block, next block of the same workgroup}
  uint32_t rec_blocks;
  uint32_t scan_grid;  // workgroups of the scan kernel
  // global tables for the big selection path
  uint32_t *g_tables;        // [n_slots * g_stride]
  uint64_t g_stride;         // words per slot
  uint32_t *g_ticket;        // work queue head (subgraph ids)
  // add_self_edge on the flat scan: per NODE of the full graph, the position in `indices` where the reference inserts its self
  // edge (ParallelSampler.cpp:386-400: lower_bound == upper_bound of v in its own row), kEmpty for a node that lists itself.
  // A property of the full graph alone: built once per sampler handle (sg_self_slot_kernel), read once per subgraph row.
  const uint32_t *self_slot;
  // ... and per ROW of the call's subgraphs ([P*cap_nodes_scr], NULL when the flat scan does not insert self edges): self_slot of the
  // row's node, written by the selection kernels beside the row records (the node id is in a register there, the table lookup one
  // more gather among a pass's indptr pairs) -- the scan's phase A reads it with the row record in ONE coalesced round trip
  // (round 6a looked the table up there: a dependent LDS -> random 4-byte HBM gather on every round's critical path).
  uint32_t *s_selfpos;
  // budgeted k-hop whose level sizes alone put every ordinary subgraph far beyond the LDS tables (depth 3, budget 20: ~4 600 nodes
  // against 2 048): the LDS attempt is not made, every subgraph goes to the global-table kernel straight away (0.07 ms per
  // 256-root call of attempts that were abandoned anyway).  A subgraph that would have fitted is merely selected over global tables.
  uint32_t lds_skip;
};

struct Tables {
  uint32_t *hkey;    // [H + kStash] bucket array (4 keys / bucket) + stash
  uint32_t *hval;    // [H + kStash] level mask of the frontier expansion
  float *pprv;       // [H + kStash] (ppr method) or nullptr
  uint32_t *nodes;   // [capn]
  uint32_t *front0;  // [capf]
  uint32_t *front1;  // [capf]
};


// ---------------------------------------------------------------- hash set
// first key slot of the bucket of `key`
__device__ __forceinline__ uint32_t bucket_base(uint32_t key, uint32_t hshift) {
  return ((key * 0x9E3779B1u) >> hshift) << 2;
}

// insert `key`; returns its slot.  New keys are appended to t.nodes.
__device__ __forceinline__ uint32_t tab_insert(const Tables &t, uint32_t *ctrl, uint32_t key,
                                               uint32_t H, uint32_t hshift, uint32_t capn) {
  const uint32_t base = bucket_base(key, hshift);
  uint32_t slot = base;
  bool is_new = false, done = false;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (!done) {
      const uint32_t old = atomicCAS(&t.hkey[base + i], kEmpty, key);
      if (old == kEmpty || old == key) { slot = base + i; is_new = (old == kEmpty); done = true; }
    }
  }
  if (!done) {
    for (uint32_t i = 0; i < kStash; i++) {
      const uint32_t old = atomicCAS(&t.hkey[H + i], kEmpty, key);
      if (old == kEmpty || old == key) {
        slot = H + i; is_new = (old == kEmpty); done = true;
        if (is_new) atomicMax(&ctrl[C_NSTASH], i + 1);
        break;
      }
    }
    if (!done) { atomicOr(&ctrl[C_OVF], 1u); return base; }   // stash full: redo on bigger tables
  }
  if (is_new) {
    const uint32_t idx = atomicAdd(&ctrl[C_NNODES], 1u);
    if (idx < capn) t.nodes[idx] = key;
    else atomicOr(&ctrl[C_OVF], 1u);
  }
  return slot;
}

__device__ __forceinline__ bool overflowed(uint32_t *ctrl) {
  return __hip_atomic_load(&ctrl[C_OVF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
}

// tab_insert with the first probe already made: `old0` = what atomicCAS(&hkey[base], kEmpty, key) returned
__device__ __forceinline__ uint32_t tab_insert_rest(const Tables &t, uint32_t *ctrl, uint32_t key, uint32_t H, uint32_t capn,
                                                    uint32_t base, uint32_t old0) {
  uint32_t slot = base;
  bool is_new = (old0 == kEmpty), done = (old0 == kEmpty || old0 == key);
#pragma unroll
  for (int i = 1; i < 4; i++) {
    if (!done) {
      const uint32_t old = atomicCAS(&t.hkey[base + i], kEmpty, key);
      if (old == kEmpty || old == key) { slot = base + i; is_new = (old == kEmpty); done = true; }
    }
  }
  if (!done) {
    for (uint32_t i = 0; i < kStash; i++) {
      const uint32_t old = atomicCAS(&t.hkey[H + i], kEmpty, key);
      if (old == kEmpty || old == key) {
        slot = H + i; is_new = (old == kEmpty); done = true;
        if (is_new) atomicMax(&ctrl[C_NSTASH], i + 1);
        break;
      }
    }
    if (!done) { atomicOr(&ctrl[C_OVF], 1u); return base; }
  }
  if (is_new) {
    const uint32_t idx = atomicAdd(&ctrl[C_NNODES], 1u);
    if (idx < capn) t.nodes[idx] = key;
    else atomicOr(&ctrl[C_OVF], 1u);
  }
  return slot;
}

// touch() for the <= 4 draws of one work item: the first probes of all keys go out TOGETHER, and so do the level-mask
// updates -- over global-memory tables every atomic is an L2 round trip, and four draws one after the other were
// eight of them in a row.  Same table contents as four touch() calls (atomics on one address are ordered: a key drawn
// twice finds itself).
__device__ __forceinline__ void touch4(const Tables &t, uint32_t *ctrl, const uint32_t (&u)[4], uint32_t cnt, uint32_t H,
                                       uint32_t hshift, uint32_t capn, uint32_t capf, bool last, uint32_t bit_next,
                                       uint32_t *nxt, int nxt_cnt_idx) {
  if (overflowed(ctrl)) return;
  uint32_t base[4], old0[4], slot[4];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    base[d] = bucket_base(u[d], hshift);
    old0[d] = kEmpty;
    if ((uint32_t)d < cnt) old0[d] = atomicCAS(&t.hkey[base[d]], kEmpty, u[d]);
  }
#pragma unroll
  for (int d = 0; d < 4; d++) {
    slot[d] = base[d];
    if ((uint32_t)d < cnt) slot[d] = tab_insert_rest(t, ctrl, u[d], H, capn, base[d], old0[d]);
  }
  if (last) return;
  uint32_t oldm[4];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    oldm[d] = bit_next;
    if ((uint32_t)d < cnt) oldm[d] = atomicOr(&t.hval[slot[d]], bit_next);
  }
#pragma unroll
  for (int d = 0; d < 4; d++) {
    if ((uint32_t)d < cnt && !(oldm[d] & bit_next)) {
      const uint32_t idx = atomicAdd(&ctrl[nxt_cnt_idx], 1u);
      if (idx < capf) nxt[idx] = u[d];
      else atomicOr(&ctrl[C_OVF], 4u);
    }
  }
}

// slot of `key` or -1.  `nstash` = number of stash entries in use.
__device__ __forceinline__ int32_t tab_find(const uint32_t *hkey, uint32_t key, uint32_t H,
                                            uint32_t hshift, uint32_t nstash) {
  const uint32_t base = bucket_base(key, hshift);
  const uint4 k4 = *reinterpret_cast<const uint4 *>(hkey + base);
  int32_t r = -1;
  r = (k4.w == key) ? (int32_t)(base + 3) : r;
  r = (k4.z == key) ? (int32_t)(base + 2) : r;
  r = (k4.y == key) ? (int32_t)(base + 1) : r;
  r = (k4.x == key) ? (int32_t)(base + 0) : r;
  if (nstash != 0 && r < 0 && k4.w != kEmpty) {
    for (uint32_t i = 0; i < nstash; i++)
      if (hkey[H + i] == key) { r = (int32_t)(H + i); break; }
  }
  return r;
}

// add `u` to the touched set and (unless this is the last level) to the next frontier
__device__ __forceinline__ void touch(const Tables &t, uint32_t *ctrl, uint32_t u, uint32_t H,
                                      uint32_t hshift, uint32_t capn, uint32_t capf, bool last,
                                      uint32_t bit_next, uint32_t *nxt, int nxt_cnt_idx) {
  if (overflowed(ctrl)) return;
  const uint32_t slot = tab_insert(t, ctrl, u, H, hshift, capn);
  if (!last) {
    const uint32_t old = atomicOr(&t.hval[slot], bit_next);
    if (!(old & bit_next)) {
      const uint32_t idx = atomicAdd(&ctrl[nxt_cnt_idx], 1u);
      if (idx < capf) nxt[idx] = u;
      else atomicOr(&ctrl[C_OVF], 4u);
    }
  }
}

// All-ascending bitonic sort of a[0..n) with virtual +inf padding (no storage
// for the padding: a compare-exchange whose upper partner is >= n is a no-op).
__device__ __forceinline__ void block_sort_u32(uint32_t *a, uint32_t n) {
  if (n < 2) { __syncthreads(); return; }
  uint32_t p2 = 1;
  while (p2 < n) p2 <<= 1;
  const uint32_t half = p2 >> 1;
  for (uint32_t k = 2; k <= p2; k <<= 1) {
    const uint32_t hk = k >> 1;
    for (uint32_t t = threadIdx.x; t < half; t += blockDim.x) {
      const uint32_t i = (t / hk) * k + (t % hk);
      const uint32_t j = i ^ (k - 1);
      if (j < n) {
        const uint32_t x = a[i], y = a[j];
        if (x > y) { a[i] = y; a[j] = x; }
      }
    }
    __syncthreads();
    for (uint32_t s = hk >> 1; s >= 1; s >>= 1) {
      for (uint32_t t = threadIdx.x; t < half; t += blockDim.x) {
        const uint32_t i = (t / s) * (2 * s) + (t % s);
        const uint32_t j = i + s;
        if (j < n) {
          const uint32_t x = a[i], y = a[j];
          if (x > y) { a[i] = y; a[j] = x; }
        }
      }
      __syncthreads();
    }
  }
}

// Ascending sort of n DISTINCT ids in LDS by counting: 256 buckets on the offset from the smallest id (sampled node ids
// spread evenly: ~n / 256 per bucket), keys scattered bucket by bucket into `tmp`, rank = bucket start + smaller keys in
// the bucket (consecutive LDS words).  Seven barriers where the bitonic network needs log2(n)^2 / 2.  `tmp` holds
// n + 768 words; ctrl words C_CHANGED / C_MFAIL / C_M are scratch.  Returns false (a left untouched) when the ids
// cluster so much that a bucket holds more than 64 of them -- the caller then sorts bitonically.
__device__ __forceinline__ bool block_sort_counting(uint32_t *a, uint32_t n, uint32_t *tmp, uint32_t *ctrl) {
  const uint32_t tid = threadIdx.x, T = blockDim.x, lane = lane_id();
  uint32_t *cnt = tmp + n, *start = cnt + 256, *cur = start + 256;
  if (tid == 0) { ctrl[C_CHANGED] = 0xFFFFFFFFu; ctrl[C_MFAIL] = 0; ctrl[C_M] = 0; }
  for (uint32_t i = tid; i < 256; i += T) cnt[i] = 0;
  __syncthreads();
  uint32_t lo = 0xFFFFFFFFu, hi = 0;
  for (uint32_t i = tid; i < n; i += T) { const uint32_t v = a[i]; lo = min(lo, v); hi = max(hi, v); }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
  }
  if (lane == 0) { atomicMin(&ctrl[C_CHANGED], lo); atomicMax(&ctrl[C_MFAIL], hi); }
  __syncthreads();
  const uint32_t vmin = ctrl[C_CHANGED], vmax = ctrl[C_MFAIL];
  uint32_t shift = 0;
  while (((vmax - vmin) >> shift) >= 256u) shift++;
  for (uint32_t i = tid; i < n; i += T) atomicAdd(&cnt[(a[i] - vmin) >> shift], 1u);
  __syncthreads();
  if (tid < 64) {                                            // exclusive scan of the 256 counts: 4 per lane of wave 0
    uint32_t c4[4], sum = 0, mx = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { c4[q] = cnt[tid * 4 + q]; sum += c4[q]; mx = max(mx, c4[q]); }
    const uint32_t incl = wave_incl_scan(sum);
    uint32_t run = incl - sum;
#pragma unroll
    for (int q = 0; q < 4; q++) { start[tid * 4 + q] = run; cur[tid * 4 + q] = run; run += c4[q]; }
    mx = wave_reduce_max(mx);
    if (tid == 0) ctrl[C_M] = mx;
  }
  __syncthreads();
  if (ctrl[C_M] > 64u) return false;
  for (uint32_t i = tid; i < n; i += T) { const uint32_t v = a[i]; tmp[atomicAdd(&cur[(v - vmin) >> shift], 1u)] = v; }
  __syncthreads();
  for (uint32_t j = tid; j < n; j += T) {
    const uint32_t v = tmp[j], b = (v - vmin) >> shift;
    const uint32_t s0 = start[b], c = cnt[b];
    uint32_t r = s0;
    for (uint32_t k = 0; k < c; k++) r += (tmp[s0 + k] < v) ? 1u : 0u;
    a[r] = v;
  }
  __syncthreads();
  return true;
}

__device__ __forceinline__ uint32_t rl_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ bool is_root(const uint32_t *roots, int R, uint32_t v) {
  bool r = false;
  for (int i = 0; i < R; i++) r |= (roots[i] == v);
  return r;
}

// ---------------------------------------------------------------------------
// Select + sort + row records of ONE subgraph `s` with the calling workgroup.
//   kGlobalTables: tables live in global memory (big path) -> L1 fences
// ---------------------------------------------------------------------------
template <bool kGlobalTables>
__device__ __forceinline__ void select_subgraph(const SampleParams &p, uint32_t s, const Tables &t,
                                                uint32_t *ctrl, uint32_t *wsum, uint32_t *lds_sort = nullptr, uint32_t lds_sort_cap = 0) {
  const uint32_t tid = threadIdx.x, T = blockDim.x;
  const uint32_t lane = lane_id(), wave = wave_id(), nw = T >> 6;
  const uint32_t H = p.H, hshift = p.hshift;
  const uint32_t capn = p.capn, capf = p.capf;
  const int R = p.R;
  const uint32_t *roots = p.roots + (size_t)s * R;
  const uint64_t serial = p.serial_base + s;
  uint32_t *res = p.s_cnt + (size_t)s * R_WORDS;

  const uint64_t tsel0 = clock64();
  // ---- phase 0: clear tables
  for (uint32_t i = tid; i < H + kStash; i += T) { t.hkey[i] = kEmpty; t.hval[i] = 0; }
  if (tid < C_WORDS) ctrl[tid] = 0;
  __syncthreads();

  // ---- phase 1: node selection
  if (p.method == SG_METHOD_PPR) {
    // ParallelSampler::ppr, .cpp:572-590 (write order: later writes win)
    for (int r = 0; r < R; r++) {
      const uint32_t root = roots[r];
      const int32_t row = p.ppr_row[root];
      const uint32_t size_all = row < 0 ? 0u : p.ppr_len[row];
      const uint32_t size_neigh = min((uint32_t)p.k, size_all);
      const uint32_t *nb = p.ppr_neigh + (size_t)(row < 0 ? 0 : row) * p.ppr_stride;
      const float *sc = p.ppr_score + (size_t)(row < 0 ? 0 : row) * p.ppr_stride;
      const float max_ppr = size_neigh > 1 ? sc[1] : 0.0f;
      if (tid == 0) {
        const uint32_t slot = tab_insert(t, ctrl, root, H, hshift, capn);
        t.pprv[slot] = (size_neigh <= 1 && size_all > 0) ? sc[0] : -1.0f;   // :574, :581
        ctrl[C_MFAIL] = size_neigh;
      }
      __syncthreads();
      // first index failing the threshold test (:584); scores are non-increasing
      for (uint32_t i = tid; i < size_neigh; i += T) {
        const bool fail = (max_ppr == 0.0f) || (sc[i] / max_ppr < p.threshold);
        if (fail) atomicMin(&ctrl[C_MFAIL], i);
      }
      __syncthreads();
      const uint32_t m = ctrl[C_MFAIL];
      for (uint32_t i = tid; i < m; i += T) {
        const uint32_t slot = tab_insert(t, ctrl, nb[i], H, hshift, capn);
        t.pprv[slot] = sc[i];                                                // :587
      }
      __syncthreads();
    }
  } else {
    // roots: level 0 (.cpp:519-522), nodeIID: roots only (.cpp:502-505)
    if (tid < (uint32_t)R) {
      const uint32_t slot = tab_insert(t, ctrl, roots[tid], H, hshift, capn);
      const uint32_t old = atomicOr(&t.hval[slot], 1u);
      if (!(old & 1u)) {
        const uint32_t idx = atomicAdd(&ctrl[C_NF0], 1u);
        t.front0[idx] = roots[tid];
      }
    }
    __syncthreads();
    const int depth = p.method == SG_METHOD_KHOP ? p.depth : 0;
    const int budget = p.budget;
    uint32_t fr_nodes = 0, fr_reads = 0;
    for (int lvl = 0; lvl < depth; lvl++) {
      const uint32_t *cur = (lvl & 1) ? t.front1 : t.front0;
      uint32_t *nxt = (lvl & 1) ? t.front0 : t.front1;
      const int cur_idx = (lvl & 1) ? C_NF1 : C_NF0, nxt_idx = (lvl & 1) ? C_NF0 : C_NF1;
      const uint32_t nf = min(ctrl[cur_idx], capf);
      __syncthreads();
      // LDS tables, a budgeted level whose draws alone are twice the table's node capacity, and a global-table pass behind
      // this kernel: hand the subgraph over right away instead of drawing until the table overflows (depth-3 k-hop: every
      // subgraph of the call; the result is the big kernel's either way -- this only saves the doomed attempt)
      if (!kGlobalTables && budget >= 0 && p.cap_nodes_scr > capn && (uint64_t)nf * (uint64_t)budget > 2ull * capn) {
        if (tid == 0) atomicOr(&ctrl[C_OVF], 1u);
        __syncthreads();
        break;
      }
      if (tid == 0) ctrl[nxt_idx] = 0;
      __syncthreads();
      const bool last = (lvl + 1 == depth);
      const uint32_t bit_next = 1u << (lvl + 1);
      if (budget >= 0) {
        // one work item = (frontier node, group of 4 draws); the 4 loads go out together
        const uint32_t groups = ((uint32_t)budget + 3u) >> 2;
        const uint32_t items = nf * groups;
        for (uint32_t q = tid; q < items; q += T) {
          const uint32_t fi = q / groups, g = q - fi * groups;
          const uint32_t v = cur[fi];
          const uint32_t e0 = p.indptr[v], deg = p.indptr[v + 1] - e0;
          const uint32_t d0 = g * 4;
          if (g == 0) { fr_nodes++; fr_reads += min(deg, (uint32_t)budget); }
          uint32_t off[4];
          uint32_t cnt4;
          if (deg <= (uint32_t)budget) {                               // .cpp:528-531
            cnt4 = d0 < deg ? min(4u, deg - d0) : 0u;
#pragma unroll
            for (int d = 0; d < 4; d++) off[d] = d0 + d;
          } else {                                                     // .cpp:533-536
            uint32_t rnd[4];
            philox4x32_10(v, (uint32_t)lvl * 65536u + g, (uint32_t)serial, (uint32_t)(serial >> 32),
                          (uint32_t)p.seed, (uint32_t)(p.seed >> 32), rnd);
            cnt4 = min(4u, (uint32_t)budget - d0);
#pragma unroll
            for (int d = 0; d < 4; d++) off[d] = __umulhi(rnd[d], deg);
          }
          uint32_t u4[4];
#pragma unroll
          for (int d = 0; d < 4; d++) u4[d] = ((uint32_t)d < cnt4) ? p.indices[e0 + off[d]] : kEmpty;
          touch4(t, ctrl, u4, cnt4, H, hshift, capn, capf, last, bit_next, nxt, nxt_idx);
        }
      } else {
        // full expansion: one wavefront streams one frontier row (coalesced)
        for (uint32_t fi = wave; fi < nf; fi += nw) {
          const uint32_t v = cur[fi];
          const uint32_t e0 = p.indptr[v], deg = p.indptr[v + 1] - e0;
          if (lane == 0) { fr_nodes++; fr_reads += deg; }
          for (uint32_t d = lane; d < deg; d += 64)
            touch(t, ctrl, p.indices[e0 + d], H, hshift, capn, capf, last, bit_next, nxt, nxt_idx);
        }
      }
      __syncthreads();
      if (overflowed(ctrl)) break;
    }
    fr_nodes = wave_reduce_sum(fr_nodes);
    fr_reads = wave_reduce_sum(fr_reads);
    if (lane == 0) { atomicAdd(&ctrl[C_FRONT_NODES], fr_nodes); atomicAdd(&ctrl[C_FRONT_READS], fr_reads); }
    __syncthreads();
  }

  // Global-memory tables: the hash set was built with L2 atomics, the phases
  // below read it with plain loads -> drop this CU's possibly stale L1 lines.
  if (kGlobalTables) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __syncthreads(); }

  const uint32_t n_all = ctrl[C_NNODES];
  if (overflowed(ctrl) || n_all > capn || n_all > p.cap_nodes_scr) {
    if (tid == 0) {
      res[R_N] = n_all; res[R_E] = 0; res[R_FLAGS] = 1u; res[R_SLOTS] = 0; res[R_Q] = 0;
      res[R_FNODES] = ctrl[C_FRONT_NODES]; res[R_FREADS] = ctrl[C_FRONT_READS];
    }
    __syncthreads();
    return;
  }
  const uint32_t n = n_all;
  const uint32_t nstash = ctrl[C_NSTASH];

  // ---- phase 2: sort ids ascending (.cpp:362); the sorted position is the sub id (.cpp:369-372)
  const uint64_t tsel1 = clock64();
  // (hval -- the expansion's level masks -- is free from here on: scratch of the counting sort when it is large enough)
  // (global tables: the ids are sorted in LDS when the kernel was given room for them -- the bitonic network over global
  //  memory cost a depth-3 subgraph of 4 600 nodes as much as its whole expansion)
  const uint32_t *sorted = t.nodes;
  if (kGlobalTables && lds_sort && n <= lds_sort_cap) {
    for (uint32_t i = tid; i < n; i += T) lds_sort[i] = t.nodes[i];
    __syncthreads();
    if (n < 64u || !block_sort_counting(lds_sort, n, lds_sort + lds_sort_cap, ctrl)) block_sort_u32(lds_sort, n);
    sorted = lds_sort;
  } else if (kGlobalTables || n < 64u || (uint64_t)n + 768u > (uint64_t)H + kStash || !block_sort_counting(t.nodes, n, t.hval, ctrl)) {
    block_sort_u32(t.nodes, n);
  }
  const uint64_t tsel2 = clock64();
  uint32_t *g_nodes = p.s_nodes + (size_t)s * p.cap_nodes_scr;
  float *g_ppr = p.s_ppr + (size_t)s * p.cap_nodes_scr;
  RowInfo *g_info = p.s_rowinfo + (size_t)s * p.cap_nodes_scr;
  uint32_t *g_rowq = p.s_rowq + (size_t)s * (p.cap_nodes_scr + 1);
  // ---- phase 3: per-row slot prefix (deg+1), quad prefix, row start -> the subgraph's row records.  The indptr pair
  //      of the NEXT pass of T rows is loaded before this pass's scans (one global round trip per pass is hidden), and
  //      the two prefixes share their barriers.
  uint32_t carry_s = 0, carry_q = 0;
  uint32_t nv = 0, ne0 = 0, ne1 = 0, nsp = 0;
  uint32_t *g_selfpos = p.s_selfpos ? p.s_selfpos + (size_t)s * p.cap_nodes_scr : nullptr;
  if (tid < n) { nv = sorted[tid]; ne0 = p.indptr[nv]; ne1 = p.indptr[nv + 1]; if (g_selfpos) nsp = p.self_slot[nv]; }
  for (uint32_t base = 0; base < n; base += T) {
    const uint32_t i = base + tid;
    const uint32_t v = nv, e0 = ne0, e1 = ne1, sp = nsp;
    if (i + T < n) { nv = sorted[i + T]; ne0 = p.indptr[nv]; ne1 = p.indptr[nv + 1]; if (g_selfpos) nsp = p.self_slot[nv]; }
    uint32_t vs = 0, vq = 0;
    if (i < n) {
      g_nodes[i] = v;
      if (p.method == SG_METHOD_PPR) {
        const int32_t slot = tab_find(t.hkey, v, H, hshift, nstash);
        g_ppr[i] = t.pprv[slot];                                         // .cpp:365
      } else {
        g_ppr[i] = -1.0f;                                                // .cpp:545
      }
      vs = e1 - e0 + 1u;
      vq = (e1 > e0) ? (((e1 - 1u) >> 2) - (e0 >> 2) + 1u) : 0u;
    }
    uint32_t tot_s, tot_q, ex_q;
    const uint32_t ex_s = block_excl_scan2(vs, vq, wsum, &tot_s, &ex_q, &tot_q);
    if (i < n) {
      RowInfo ri;
      ri.e0 = e0; ri.deg = vs - 1u; ri.rs = carry_s + ex_s; ri.v = v;
      g_info[i] = ri;
      g_rowq[i] = carry_q + ex_q;
      if (g_selfpos) g_selfpos[i] = sp;
    }
    carry_s += tot_s; carry_q += tot_q;
  }
  if (tid == 0) g_rowq[n] = carry_q;
  if (tid < (uint32_t)R) {                                             // .cpp:373-377: sub id of the root(s)
    const uint32_t root = roots[tid];
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted[mid] < root) lo = mid + 1; else hi = mid; }
    p.s_tgt[(size_t)s * kMaxRoots + tid] = lo;
  }
  if (tid == 0) {
    res[R_N] = n; res[R_E] = 0; res[R_FLAGS] = 0; res[R_SLOTS] = carry_s; res[R_Q] = carry_q;
    res[R_FNODES] = ctrl[C_FRONT_NODES]; res[R_FREADS] = ctrl[C_FRONT_READS];
    // phase cycles (units of 16) of the selection, summed over the call's subgraphs (sg_debug_scan_phases words 20..22)
    const uint64_t tsel3 = clock64();
    atomicAdd(&p.plan[20], (uint32_t)((tsel1 - tsel0) >> 4));
    atomicAdd(&p.plan[21], (uint32_t)((tsel2 - tsel1) >> 4));
    atomicAdd(&p.plan[22], (uint32_t)((tsel3 - tsel2) >> 4));
  }
  __syncthreads();
}

// LDS carve shared by host (size computation) and device
struct LdsLayout {
  size_t hkey, hval, pprv, nodes, front0, front1, ctrl, wsum, total;
};

__host__ __device__ inline size_t r16(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline LdsLayout lds_layout(uint32_t H, uint32_t capn, uint32_t capf, bool ppr) {
  LdsLayout L;
  size_t o = 0;
  L.hkey = o; o += r16((size_t)(H + kStash) * 4);
  L.hval = o; o += r16((size_t)(H + kStash) * 4);
  L.pprv = o; o += ppr ? r16((size_t)(H + kStash) * 4) : 0;
  L.nodes = o; o += r16((size_t)capn * 4);
  L.front0 = o; o += r16((size_t)capf * 4);
  L.front1 = o; o += r16((size_t)capf * 4);
  L.ctrl = o; o += C_WORDS * 4;
  L.wsum = o; o += 32 * 4;
  L.total = o;
  return L;
}

__global__ void sg_select_lds_kernel(SampleParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_next;
  const LdsLayout L = lds_layout(p.H, p.capn, p.capf, p.method == SG_METHOD_PPR);
  Tables t;
  t.hkey = (uint32_t *)(smem + L.hkey);
  t.hval = (uint32_t *)(smem + L.hval);
  t.pprv = (float *)(smem + L.pprv);
  t.nodes = (uint32_t *)(smem + L.nodes);
  t.front0 = (uint32_t *)(smem + L.front0);
  t.front1 = (uint32_t *)(smem + L.front1);
  uint32_t *ctrl = (uint32_t *)(smem + L.ctrl);
  uint32_t *wsum = (uint32_t *)(smem + L.wsum);
  // persistent workgroups: subgraph ids come from a global ticket
  for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(p.g_ticket, 1u);
    __syncthreads();
    const uint32_t s = s_next;
    __syncthreads();
    if (s >= p.P) return;
    if (p.lds_skip) {
      if (threadIdx.x == 0) {
        uint32_t *res = p.s_cnt + (size_t)s * R_WORDS;
        res[R_N] = 0; res[R_E] = 0; res[R_FLAGS] = 1u; res[R_SLOTS] = 0; res[R_Q] = 0; res[R_FNODES] = 0; res[R_FREADS] = 0;
      }
      continue;
    }
    select_subgraph<false>(p, s, t, ctrl, wsum);
  }
}

// Big path: persistent workgroups pull overflowed subgraphs (flag bit0 from the
// LDS kernel) from a ticket counter and redo them over global-memory tables.
// Dynamic LDS (optional): [2 * lds_sort_cap + 768] words -- the node ids are sorted there (p.capm carries lds_sort_cap for
// this launch; 0 = no room: ids sorted in global memory).
__global__ void sg_select_big_kernel(SampleParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_big[];
  __shared__ uint32_t ctrl[C_WORDS];
  __shared__ uint32_t wsum[32];
  __shared__ uint32_t s_next;
  uint32_t *base = p.g_tables + (size_t)blockIdx.x * p.g_stride;
  Tables t;
  size_t o = 0;
  const size_t hs = (size_t)p.H + kStash;
  t.hkey = base + o; o += hs;
  t.hval = base + o; o += hs;
  t.pprv = (float *)(base + o); o += hs;
  t.nodes = base + o; o += (p.capn + 3u) & ~3u;
  t.front0 = base + o; o += (p.capf + 3u) & ~3u;
  t.front1 = base + o; o += (p.capf + 3u) & ~3u;
  for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(p.g_ticket, 1u);
    __syncthreads();
    const uint32_t s = s_next;
    __syncthreads();
    if (s >= p.P) return;
    const uint32_t flags = p.s_cnt[(size_t)s * R_WORDS + R_FLAGS];
    if (!(flags & 1u)) continue;
    select_subgraph<true>(p, s, t, ctrl, wsum, p.capm ? reinterpret_cast<uint32_t *>(smem_big) : nullptr, p.capm);
  }
}

}  // namespace shadow
