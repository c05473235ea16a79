// sampler_device.h -- device side of the MI355X subgraph sampler.
//
// One WORKGROUP samples one subgraph (persistent workgroups pull subgraph ids
// from a ticket).  Phases, all inside one kernel:
//
//   1. node selection: k-hop frontier expansion (ParallelSampler.cpp:510-547),
//      PPR top-k (.cpp:565-590) or the roots alone (.cpp:498-505).  Dedupe goes
//      through an LDS hash set of 16-byte buckets (4 keys per bucket: a lookup is
//      ONE ds_read_b128 and four compares, no probe loop) with a tiny overflow
//      stash; frontiers are LDS lists; budgeted draws are Philox4x32-10.
//   2. ids sorted ascending in LDS (== std::sort, .cpp:362); the hash value of
//      each id becomes its rank (orig2subID, .cpp:369-372).
//   3. per row: slot prefix (deg+1 slots: position keys), quad prefix (aligned
//      16-byte quads of the row in the indices array) and row start.
//   4. node-induced slicing (.cpp:378-431) as an ORDER-FREE streaming scan of the
//      quads: every lane loads one aligned uint4 of neighbour ids (partial quads
//      at row ends are masked), looks the four ids up, and appends the rare
//      matches to an LDS list keyed by 2*slot+kind.  Waves pull chunks of quads
//      from an LDS ticket; up to kUnroll quad groups (1 KiB each) are in flight
//      per wave.  Trailing self edges, empty rows and the compat over-read are
//      emitted by a per-row pre-pass.
//   5. the match list is bucket-sorted by key (== the reference's edge order)
//      and written to the subgraph's scratch CSR.
//
// The same code is instantiated over global-memory tables for subgraphs whose
// node set exceeds the LDS tables (sg_sample_big_kernel).
#pragma once
#include "common.h"

namespace shadow {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kUnroll = 4;               // quad groups (64 x 16 B) in flight per wave
constexpr uint32_t kQChunk = 1024;       // quads a wave takes per ticket (= 4096 neighbour ids)
constexpr uint32_t kLdsCapNodes = 2048;  // largest node set handled by the LDS kernel
constexpr uint32_t kMaxRoots = 8;
constexpr uint32_t kStash = 64;          // overflow entries behind the bucket array
constexpr uint32_t kSortBuckets = 256;
constexpr uint32_t kBitWords = 2048;     // membership filter: 64 Ki bits indexed by the low id bits
constexpr uint32_t kBitWordsBig = 32768; // same for the global-table path (node sets of 10^3..10^5): 1 Mi bits

// control words (LDS)
enum { C_NNODES = 0, C_NF0 = 1, C_NF1 = 2, C_OVF = 3, C_MFAIL = 4, C_FRONT_NODES = 5,
       C_FRONT_READS = 6, C_CHANGED = 7, C_M = 8, C_TICKET = 9, C_NSTASH = 10, C_MV = 11, C_WORDS = 16 };

// per-subgraph result words in scratch (s_cnt)
enum { R_N = 0, R_E = 1, R_FLAGS = 2, R_SLOTS = 3, R_FNODES = 4, R_FREADS = 5, R_T0 = 8, R_WORDS = 16 };
#ifdef SHADOW_SG_TIMING
#define SHD_STAMP(i) do { if (threadIdx.x == 0) res[R_T0 + (i)] = (uint32_t)(clock64() - t_begin); } while (0)
// fine-grained scan timers (wave 0 only): SHD_T(k) adds the cycles since the previous stamp to bucket k
#define SHD_TT(k) do { if (threadIdx.x == 0) { const uint64_t now_ = clock64(); tacc[k] += (uint32_t)(now_ - tlast); tlast = now_; } } while (0)
#if SHADOW_SG_TIMING == 2      // buckets over the generic path instead of the bulk path
#define SHD_T(k) do {} while (0)
#define SHD_G(k) SHD_TT(k)
#else
#define SHD_T(k) SHD_TT(k)
#define SHD_G(k) do {} while (0)
#endif
#else
#define SHD_STAMP(i) do {} while (0)
#define SHD_T(k) do {} while (0)
#define SHD_G(k) do {} while (0)
#endif

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
  // table geometry
  uint32_t capn;      // node capacity of the tables used by this launch
  uint32_t capf;      // frontier list capacity
  uint32_t H;         // hash key slots in buckets (multiple of 4, power of two)
  uint32_t hshift;    // 32 - log2(H/4)
  uint32_t capm;      // capacity of the match list
  // per-subgraph scratch (stride = cap_nodes_scr / cap_edges_scr)
  uint32_t cap_nodes_scr, cap_edges_scr;
  uint32_t *s_nodes;   // [P*cap_nodes_scr]
  float *s_ppr;        // [P*cap_nodes_scr]
  uint32_t *s_row;     // [P*cap_edges_scr] local row of each emitted edge
  uint32_t *s_col;     // [P*cap_edges_scr]
  uint32_t *s_eid;     // [P*cap_edges_scr]
  uint32_t *s_tgt;     // [P*kMaxRoots]
  uint32_t *s_cnt;     // [P*R_WORDS]
  // global tables for the big path
  uint32_t *g_tables;        // [n_slots * g_stride]
  uint64_t g_stride;         // words per slot
  uint32_t *g_ticket;        // work queue head (subgraph ids)
};

struct Tables {
  uint32_t *hkey;    // [H + kStash] bucket array (4 keys / bucket) + stash
  uint32_t *hval;    // [H + kStash] level mask, later the sub id
  float *pprv;       // [H + kStash] (ppr method) or nullptr
  uint32_t *nodes;   // [capn]
  uint32_t *rowptr;  // [capn+1] slot prefix (deg+1 per row)
  uint32_t *qptr;    // [capn+1] quad prefix
  uint32_t *rowe0;   // [capn] full-graph row start of each node
  uint32_t *front0;  // [capf]
  uint32_t *front1;  // [capf]
  uint32_t *lkey;    // [capm] match list: 2*slot+kind
  uint32_t *lval;    // [capm] match list: column sub id
  uint32_t *lnext;   // [capm] bucket chains of the final sort
  uint32_t *bhead;   // [kSortBuckets]
  uint32_t *bcnt;    // [kSortBuckets]
  unsigned char *wtmp;  // [waves * 128] wave-private scratch of the scan (LDS)
  uint32_t *bits;    // [kBitWords / kBitWordsBig] membership filter over the node set (LDS; aliases the frontiers)
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

__device__ __forceinline__ bool overflowed(uint32_t *ctrl) {
  return __hip_atomic_load(&ctrl[C_OVF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
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

// membership-filter probe of one aligned quad: bit c = component c may be in the node set
template <uint32_t kBW>
__device__ __forceinline__ uint32_t probe_quad(const uint32_t *bits, const uint4 c) {
  const uint32_t w0 = bits[(c.x >> 5) & (kBW - 1u)];
  const uint32_t w1 = bits[(c.y >> 5) & (kBW - 1u)];
  const uint32_t w2 = bits[(c.z >> 5) & (kBW - 1u)];
  const uint32_t w3 = bits[(c.w >> 5) & (kBW - 1u)];
  return ((w0 >> (c.x & 31u)) & 1u) | (((w1 >> (c.y & 31u)) & 1u) << 1) |
         (((w2 >> (c.z & 31u)) & 1u) << 2) | (((w3 >> (c.w & 31u)) & 1u) << 3);
}

__device__ __forceinline__ uint32_t rl_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ bool is_root(const uint32_t *roots, int R, uint32_t v) {
  bool r = false;
  for (int i = 0; i < R; i++) r |= (roots[i] == v);
  return r;
}

// Append the calling wave's matches to the (unordered) match list.  Each lane
// offers up to kN candidate entries; bit q of `mask` says entry q is real.
// (fixed positions + a bit mask keep the arrays in registers)
template <int kN>
__device__ __forceinline__ void emit_list(const Tables &t, uint32_t *ctrl, uint32_t capm,
                                          uint32_t mask, const uint32_t (&keys)[kN],
                                          const uint32_t (&vals)[kN]) {
  const uint64_t any = __ballot(mask != 0);
  if (any == 0) return;
  // exclusive prefix of the per-lane entry counts from kN ballots (scalar
  // popcounts + v_mbcnt; no cross-lane LDS traffic)
  const uint64_t lt = lanemask_lt();
  uint32_t excl = 0, total = 0;
#pragma unroll
  for (int q = 0; q < kN; q++) {
    const uint64_t mq = __ballot((mask >> q) & 1u);
    excl += __popcll(mq & lt);
    total += __popcll(mq);
  }
  uint32_t base = 0;
  if (lane_id() == 0) base = atomicAdd(&ctrl[C_M], total);
  base = __builtin_amdgcn_readfirstlane(base);
  const uint32_t r0 = base + excl;
#pragma unroll
  for (int q = 0; q < kN; q++) {
    if ((mask >> q) & 1u) {
      const uint32_t r = r0 + __popc(mask & ((1u << q) - 1u));
      if (r < capm) { t.lkey[r] = keys[q]; t.lval[r] = vals[q]; }
    }
  }
}

// Scan variant: entry q of a lane has key keybase+q (q even: self edge before
// component q/2 -> value myrow, q odd: regular edge of component q/2 -> value cols[q/2]).
__device__ __forceinline__ void emit_scan(const Tables &t, uint32_t *ctrl, uint32_t capm,
                                          uint32_t mask, uint32_t keybase, uint32_t myrow,
                                          const uint32_t (&cols)[4]) {
  const uint64_t any = __ballot(mask != 0);
  if (any == 0) return;
  const uint64_t lt = lanemask_lt();
  uint32_t excl = 0, total = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const uint64_t mq = __ballot((mask >> q) & 1u);
    if (mq) { excl += __popcll(mq & lt); total += __popcll(mq); }
  }
  uint32_t base = 0;
  if (lane_id() == 0) base = atomicAdd(&ctrl[C_M], total);
  base = __builtin_amdgcn_readfirstlane(base);
  const uint32_t r0 = base + excl;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    if ((mask >> q) & 1u) {
      const uint32_t r = r0 + __popc(mask & ((1u << q) - 1u));
      if (r < capm) { t.lkey[r] = keybase + q; t.lval[r] = (q & 1) ? cols[q >> 1] : myrow; }
    }
  }
}

// ---------------------------------------------------------------------------
// Sample ONE subgraph `s` with the calling workgroup.
//   kGlobalTables: tables live in global memory (big path) -> L1 fences
//   kPlain: no self-edge insertion, no compat over-read, no root-root exclusion
// ---------------------------------------------------------------------------
template <bool kGlobalTables, bool kPlain>
__device__ __forceinline__ void sample_subgraph(const SampleParams &p, uint32_t s, const Tables &t,
                                                uint32_t *ctrl, uint32_t *wsum) {
  const uint32_t tid = threadIdx.x, T = blockDim.x;
  const uint32_t lane = lane_id(), wave = wave_id(), nw = T >> 6;
  const uint32_t H = p.H, hshift = p.hshift;
  const uint32_t capn = p.capn, capf = p.capf;
  const int R = p.R;
  const uint32_t *roots = p.roots + (size_t)s * R;
  const uint64_t serial = p.serial_base + s;
  uint32_t *res = p.s_cnt + (size_t)s * R_WORDS;
#ifdef SHADOW_SG_TIMING
  const uint64_t t_begin = clock64();
  uint64_t tlast = t_begin;
  uint32_t tacc[4] = {0, 0, 0, 0};
#endif

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
#pragma unroll
          for (int d = 0; d < 4; d++)
            if ((uint32_t)d < cnt4) touch(t, ctrl, u4[d], H, hshift, capn, capf, last, bit_next, nxt, nxt_idx);
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

  SHD_STAMP(0);   // selection done
  const uint32_t n_all = ctrl[C_NNODES];
  if (overflowed(ctrl) || n_all > capn || n_all > p.cap_nodes_scr) {
    if (tid == 0) {
      res[R_N] = n_all; res[R_E] = 0; res[R_FLAGS] = 1u; res[R_SLOTS] = 0;
      res[R_FNODES] = ctrl[C_FRONT_NODES]; res[R_FREADS] = ctrl[C_FRONT_READS];
    }
    __syncthreads();
    return;
  }
  const uint32_t n = n_all;
  const uint32_t nstash = ctrl[C_NSTASH];

  // ---- phase 2: sort ids ascending (.cpp:362) and rank them (.cpp:369-372)
  block_sort_u32(t.nodes, n);
  uint32_t *g_nodes = p.s_nodes + (size_t)s * p.cap_nodes_scr;
  float *g_ppr = p.s_ppr + (size_t)s * p.cap_nodes_scr;
  // ---- phase 3 (fused): per-row slot prefix (deg+1), quad prefix, row start
  uint32_t carry_s = 0, carry_q = 0;
  for (uint32_t base = 0; base < n; base += T) {
    const uint32_t i = base + tid;
    uint32_t vs = 0, vq = 0;
    if (i < n) {
      const uint32_t v = t.nodes[i];
      const int32_t slot = tab_find(t.hkey, v, H, hshift, nstash);
      t.hval[slot] = i;
      g_nodes[i] = v;
      g_ppr[i] = (p.method == SG_METHOD_PPR) ? t.pprv[slot] : -1.0f;     // .cpp:365, :545
      const uint32_t e0 = p.indptr[v], e1 = p.indptr[v + 1];
      t.rowe0[i] = e0;
      vs = e1 - e0 + 1u;
      vq = (e1 > e0) ? (((e1 - 1u) >> 2) - (e0 >> 2) + 1u) : 0u;
    }
    uint32_t tot_s, tot_q;
    const uint32_t ex_s = block_excl_scan(vs, wsum, &tot_s);
    const uint32_t ex_q = block_excl_scan(vq, wsum, &tot_q);
    if (i < n) { t.rowptr[i] = carry_s + ex_s; t.qptr[i] = carry_q + ex_q; }
    carry_s += tot_s; carry_q += tot_q;
  }
  if (tid == 0) { t.rowptr[n] = carry_s; t.qptr[n] = carry_q; }
  __syncthreads();
  if (tid < (uint32_t)R) {                                             // .cpp:373-377
    const int32_t slot = tab_find(t.hkey, roots[tid], H, hshift, nstash);
    p.s_tgt[(size_t)s * kMaxRoots + tid] = t.hval[slot];
  }
  const uint32_t S = carry_s, Q = carry_q;
  // membership filter for the streaming scan: bit (id mod 2^16) of every node.  A set bit is only a
  // candidate (resolved exactly against the hash table after the scan); a clear bit is a definite miss.
  constexpr uint32_t kBW = kGlobalTables ? kBitWordsBig : kBitWords;
  for (uint32_t w = tid; w < kBW; w += T) t.bits[w] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += T) {
    const uint32_t v = t.nodes[i];
    atomicOr(&t.bits[(v >> 5) & (kBW - 1u)], 1u << (v & 31u));
  }
  __syncthreads();
  SHD_STAMP(1);   // sort + rank + prefixes done

  // ---- phase 4: streaming induction (.cpp:381-427)
  unsigned char *wflag = t.wtmp + wave * 128u;      // wave-private: start flags per position
  unsigned char *wsel = wflag + 64;                 //               k-th starting row -> lane
  for (uint32_t i = tid; i < nw * 32u; i += T) reinterpret_cast<uint32_t *>(t.wtmp)[i] = 0;
  const bool incl_self = !kPlain && (p.include_self != 0);
  const bool itc = kPlain || (p.include_target_conn != 0) || (R == 1);   // .cpp:356-358
  const bool compat = !kPlain && (p.compat != 0);
  const uint32_t cape = p.cap_edges_scr;
  const uint32_t capm = p.capm;
  uint32_t *g_row = p.s_row + (size_t)s * cape;
  uint32_t *g_col = p.s_col + (size_t)s * cape;
  uint32_t *g_eid = p.s_eid + (size_t)s * cape;
  uint32_t e_run = 0;
  uint32_t rq0 = 0;          // quads [rq0, rq1) are scanned in this round
  uint32_t rquads = Q;
  for (;;) {
    const uint32_t rq1 = min(Q, rq0 + rquads);
    if (tid == 0) { ctrl[C_M] = 0; ctrl[C_TICKET] = 0; }
    __syncthreads();
    // ---- per-row pre-pass: trailing self edge, empty rows, compat over-read --
    //      everything that lives in the sentinel slot.  A row is handled in the
    //      round that scans its last quad (empty rows: the round holding their
    //      quad position), so every round's keys stay above the previous round's.
    if (!kPlain && (incl_self || compat)) {
      for (uint32_t base = 0; base < n; base += T) {
        const uint32_t i = base + tid;
        uint32_t keys[2] = {0, 0}, vals[2] = {0, 0}, mask = 0;
        bool mine = false;
        if (i < n) {
          const uint32_t qs = t.qptr[i], qe = t.qptr[i + 1];
          if (qe > qs) mine = (qe - 1u >= rq0 && qe - 1u < rq1);
          else mine = (qs >= rq0 && qs < rq1) || (qs == Q && rq1 == Q && (rq0 < rq1 || Q == 0 || rq0 == Q));
        }
        if (mine) {
          const uint32_t v = t.nodes[i];
          const uint32_t e0 = t.rowe0[i];
          const uint32_t rs = t.rowptr[i];
          const uint32_t deg = t.rowptr[i + 1] - rs - 1u;
          bool trailing = false;
          if (incl_self) {
            trailing = (deg == 0) || (p.indices[e0 + deg - 1] < v);     // .cpp:387-400, self goes last
            if (trailing) { keys[0] = 2u * (rs + deg); vals[0] = i; mask |= 1u; }
          }
          if (compat && !trailing) {
            bool inserted = false;
            if (incl_self) {
              // was the self edge inserted inside the row?  <=> v is not a neighbour
              uint32_t l3 = 0, h3 = deg;
              while (l3 < h3) { const uint32_t m3 = (l3 + h3) >> 1; if (p.indices[e0 + m3] < v) l3 = m3 + 1; else h3 = m3; }
              inserted = !(l3 < deg && p.indices[e0 + l3] == v);
            }
            if (!inserted && (uint64_t)e0 + deg < p.nnz) {              // .cpp:401-405
              // (a neighbour candidate like any other: resolved and root-filtered after the scan)
              const uint32_t c = p.indices[e0 + deg];
              if ((t.bits[(c >> 5) & (kBW - 1u)] >> (c & 31u)) & 1u) {
                keys[1] = 2u * (rs + deg) + 1u; vals[1] = c; mask |= 2u;
              }
            }
          }
        }
        emit_list<2>(t, ctrl, capm, mask, keys, vals);
      }
    }
    // ---- the quad scan: stream every row's aligned 16-B quads, test each id against the membership
    //      filter (one LDS dword per id, no branches) and append the rare candidates to the list:
    //      key = 2*slot + kind; kind 1: neighbour candidate (value = its global id, resolved after the
    //      scan), kind 0: inserted self edge (value = row).
    for (;;) {
      uint32_t tk = 0;
      if (lane == 0) tk = atomicAdd(&ctrl[C_TICKET], 1u);
      tk = __builtin_amdgcn_readfirstlane(tk);
      const uint32_t qa = rq0 + tk * kQChunk;
      if (qa >= rq1) break;
      const uint32_t qb = min(qa + kQChunk, rq1);
      // first row whose quads end after qa (uniform binary search)
      uint32_t lo = 0, hi = n;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (t.qptr[mid + 1] > qa) hi = mid; else lo = mid + 1;
      }
      uint32_t row = lo, qpos = qa;   // wave-uniform walk state
      while (qpos < qb) {
        // ---------------------------------------------------------------
        // bulk path (plain variant): >= kUnroll*64 INTERIOR quads of one long row -- every
        // component valid, one set of row scalars for the whole run, kUnroll 1-KiB loads in flight.
        // ---------------------------------------------------------------
        if (kPlain) {
          const uint32_t qs_row = __builtin_amdgcn_readfirstlane(t.qptr[row]);
          const uint32_t qe_row = __builtin_amdgcn_readfirstlane(t.qptr[row + 1]);
          const uint32_t e0 = __builtin_amdgcn_readfirstlane(t.rowe0[row]);
          const uint32_t rs = __builtin_amdgcn_readfirstlane(t.rowptr[row]);
          const uint32_t deg = __builtin_amdgcn_readfirstlane(t.rowptr[row + 1]) - rs - 1u;
          // interior quads: all four ids inside [e0, e0+deg)
          const uint32_t q_first = qs_row + ((e0 & 3u) ? 1u : 0u);
          const uint32_t q_last = qe_row - ((((e0 + deg) & 3u) && qe_row > qs_row) ? 1u : 0u);   // exclusive
          const uint32_t lo_q = max(qpos, q_first), hi_q = min(min(q_last, qe_row), qb);
          if (lo_q == qpos && hi_q > lo_q && hi_q - lo_q >= 64u * kUnroll) {
            const uint32_t nrun = (hi_q - lo_q) / (64u * kUnroll);
            for (uint32_t it = 0; it < nrun; it++) {
              const uint32_t qbase = qpos + it * 64u * kUnroll;
              uint4 c4[kUnroll];
              SHD_T(0);
#pragma unroll
              for (int u = 0; u < kUnroll; u++) {
                const uint32_t q = qbase + u * 64u + lane;
                const uint32_t addr4 = ((e0 >> 2) + (q - qs_row)) << 2;
                c4[u] = *reinterpret_cast<const uint4 *>(p.indices + addr4);
              }
#ifdef SHADOW_SG_TIMING
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
              SHD_T(1);
              uint32_t hit = 0;        // bit (4u + c): component c of group u is a candidate
#pragma unroll
              for (int u = 0; u < kUnroll; u++) hit |= probe_quad<kBW>(t.bits, c4[u]) << (4 * u);
              SHD_T(2);
              if (hit) {
                uint32_t r = atomicAdd(&ctrl[C_M], (uint32_t)__popc(hit));
#pragma unroll
                for (int u = 0; u < kUnroll; u++) {
                  const uint32_t cc[4] = {c4[u].x, c4[u].y, c4[u].z, c4[u].w};
                  const uint32_t q = qbase + u * 64u + lane;
                  const uint32_t j0 = (((e0 >> 2) + (q - qs_row)) << 2) - e0;
#pragma unroll
                  for (int c = 0; c < 4; c++) {
                    if ((hit >> (4 * u + c)) & 1u) {
                      if (r < capm) { t.lkey[r] = 2u * (rs + j0 + c) + 1u; t.lval[r] = cc[c]; }   // .cpp:420-422
                      r++;
                    }
                  }
                }
              }
              SHD_T(3);
            }
            qpos += nrun * 64u * kUnroll;
            // (usually the row is not finished: its last quads go through the generic path)
            while (row < n && __builtin_amdgcn_readfirstlane(t.qptr[row + 1]) <= qpos) row++;
            continue;
          }
        }
        SHD_G(0);
        uint32_t d_take[kUnroll], l_row[kUnroll], l_addr[kUnroll], l_e0[kUnroll], l_rs[kUnroll],
            l_deg[kUnroll], prevv[kUnroll];
        uint4 cand[kUnroll];
        // ---- build up to kUnroll groups of <=64 quads and put their loads in flight
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
          d_take[u] = 0; l_row[u] = 0; l_addr[u] = 0; l_e0[u] = 0; l_rs[u] = 0; l_deg[u] = 0; prevv[u] = 0;
          cand[u] = make_uint4(kEmpty, kEmpty, kEmpty, kEmpty);
          if (qpos < qb) {
            // Position -> row for the next <=64 quads without a search: lane l looks at row
            // `row + l`; non-empty rows flag the position their first quad lands on; the flags
            // read back as one ballot give every position the count of rows starting at or
            // before it; a wave-private byte table turns that count into the lane (= row offset).
            const uint32_t r_l = min(row + lane, n);                   // qptr[n] = Q
            const uint32_t qs_l = t.qptr[r_l];
            const uint32_t qe_l = t.qptr[min(r_l + 1u, n)];
            const uint32_t wend = __builtin_amdgcn_readlane(qe_l, 63);
            const uint32_t take = min(min(64u, qb - qpos), wend - qpos);
            const uint32_t rel = (lane == 0) ? 0u : qs_l - qpos;       // row `row` holds qpos
            const bool instart = (qe_l > qs_l) && (lane == 0 || rel < take);
            const uint64_t nb = __ballot(instart);
            if (instart) { wflag[rel] = 1; wsel[__popcll(nb & lanemask_lt())] = (unsigned char)lane; }
            __builtin_amdgcn_wave_barrier();
            const uint32_t fl = wflag[lane];
            __builtin_amdgcn_wave_barrier();
            wflag[lane] = 0;
            const uint64_t m64 = __ballot(fl != 0);
            const uint32_t ppos = min(lane, take - 1u);                 // lanes >= take mirror the last quad
            const uint32_t cnt = __popcll(m64 & (~0ull >> (63u - ppos)));   // >= 1: bit 0 is always set
            const uint32_t myrow = row + wsel[cnt - 1u];
            __builtin_amdgcn_wave_barrier();
            const uint32_t q = qpos + ppos;
            const uint32_t e0 = t.rowe0[myrow];
            const uint32_t rs = t.rowptr[myrow];
            const uint32_t addr4 = ((e0 >> 2) + (q - t.qptr[myrow])) << 2;   // aligned quad
            d_take[u] = take; l_row[u] = myrow; l_addr[u] = addr4; l_e0[u] = e0; l_rs[u] = rs;
            l_deg[u] = t.rowptr[myrow + 1] - rs - 1u;
            cand[u] = *reinterpret_cast<const uint4 *>(p.indices + addr4);
            if (!kPlain && incl_self && addr4 > e0) prevv[u] = p.indices[addr4 - 1];
            qpos += take;
            if (qpos < qb) {
              const uint64_t cont = __ballot(r_l < n && qe_l > qpos);   // rows ending after the new position
              row = cont ? row + (uint32_t)__ffsll((unsigned long long)cont) - 1u : row + 64u;
              while (row < n && __builtin_amdgcn_readfirstlane(t.qptr[row + 1]) <= qpos) row++;   // (only after a run of empty rows)
            }
          }
        }
        SHD_G(1);
#if SHADOW_SG_TIMING == 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        SHD_G(2);
        // ---- filter probes (branch-free, all LDS reads of the kUnroll groups in flight together)
        uint32_t hit = 0;              // bit (4u + c): neighbour candidate
        uint32_t selfm = 0;            // bit (4u + c): self edge goes right before component c
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
          const uint32_t deg = l_deg[u];
          const uint32_t j0 = l_addr[u] - l_e0[u];                 // wraps when the quad starts before the row
          uint32_t vmask = 0;
#pragma unroll
          for (int c = 0; c < 4; c++)
            if (lane < d_take[u] && (j0 + c < deg)) vmask |= 1u << c;
          hit |= (probe_quad<kBW>(t.bits, cand[u]) & vmask) << (4 * u);
          if (!kPlain && incl_self) {
            // the self edge goes right before the first neighbour > v (.cpp:387-400, :408-410)
            const uint32_t v = t.nodes[l_row[u]];
            const uint32_t cc[4] = {cand[u].x, cand[u].y, cand[u].z, cand[u].w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
              const uint32_t j = j0 + c;
              const uint32_t pv = (c == 0) ? prevv[u] : cc[c > 0 ? c - 1 : 0];
              const bool prev_lt = (j == 0) || (pv < v);
              if (((vmask >> c) & 1u) && prev_lt && v < cc[c]) selfm |= 1u << (4 * u + c);
            }
          }
        }
        if (hit | selfm) {
          uint32_t r = atomicAdd(&ctrl[C_M], (uint32_t)(__popc(hit) + __popc(selfm)));
#pragma unroll
          for (int u = 0; u < kUnroll; u++) {
            const uint32_t cc[4] = {cand[u].x, cand[u].y, cand[u].z, cand[u].w};
            const uint32_t keybase = 2u * (l_rs[u] + (l_addr[u] - l_e0[u]));
#pragma unroll
            for (int c = 0; c < 4; c++) {
              if (!kPlain && ((selfm >> (4 * u + c)) & 1u)) {
                if (r < capm) { t.lkey[r] = keybase + 2u * c; t.lval[r] = l_row[u]; }       // .cpp:408-410
                r++;
              }
              if ((hit >> (4 * u + c)) & 1u) {
                if (r < capm) { t.lkey[r] = keybase + 2u * c + 1u; t.lval[r] = cc[c]; }      // .cpp:420-422
                r++;
              }
            }
          }
        }
        SHD_G(3);
      }
    }
    __syncthreads();
    if (kGlobalTables) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __syncthreads(); }
    const uint32_t m = ctrl[C_M];
    SHD_STAMP(2);   // scan done (last round)
    if (m > capm) {
      // too many matches for the list: redo this round on half the quads
      rquads = max((rq1 - rq0) / 2, 64u);
      if (rq1 - rq0 <= 64u) {                       // cannot shrink further: list is too small
        if (tid == 0) { res[R_N] = n; res[R_E] = 0; res[R_FLAGS] = 1u; res[R_SLOTS] = S;
                        res[R_FNODES] = ctrl[C_FRONT_NODES]; res[R_FREADS] = ctrl[C_FRONT_READS]; }
        __syncthreads();
        return;
      }
      __syncthreads();
      continue;
    }
    // ---- order the matches: bucket by the high key bits, rank inside the bucket chain
    {
      uint32_t kbits = 1;
      while (kbits < 32 && (2u * S + 2u) >> kbits) kbits++;
      const uint32_t bshift = kbits > 8 ? kbits - 8 : 0;
      for (uint32_t i = tid; i < kSortBuckets; i += T) { t.bhead[i] = kEmpty; t.bcnt[i] = 0; }
      __syncthreads();
      // resolve the neighbour candidates exactly (.cpp:412-413) and chain the survivors
      for (uint32_t i = tid; i < m; i += T) {
        uint32_t key = t.lkey[i];
        if (key & 1u) {
          const uint32_t c = t.lval[i];
          const int32_t hs = tab_find(t.hkey, c, H, hshift, nstash);
          bool keep = hs >= 0;
          if (keep && !itc && is_root(roots, R, c)) {
            // multi-root subgraph without include_target_conn: drop root<->root edges (.cpp:414-418)
            const uint32_t slot = key >> 1;
            uint32_t lo = 0, hi = n;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (t.rowptr[mid] <= slot) lo = mid; else hi = mid; }
            keep = !is_root(roots, R, t.nodes[lo]);
          }
          if (keep) t.lval[i] = t.hval[hs];
          else { key = kEmpty; t.lkey[i] = kEmpty; }
        }
        if (key != kEmpty) {
          const uint32_t b = min(key >> bshift, kSortBuckets - 1);
          atomicAdd(&t.bcnt[b], 1u);
          t.lnext[i] = atomicExch(&t.bhead[b], i);
        }
      }
      __syncthreads();
      // exclusive scan of the bucket counts (kSortBuckets = 256 = 4 per lane of wave 0)
      if (wave == 0) {
        uint32_t c4[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { c4[q] = t.bcnt[lane * 4 + q]; sum += c4[q]; }
        const uint32_t incl = wave_incl_scan(sum);
        uint32_t run = incl - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) { t.bcnt[lane * 4 + q] = run; run += c4[q]; }
        if (lane == 63) ctrl[C_MV] = run;          // surviving entries
      }
      __syncthreads();
      for (uint32_t i = tid; i < m; i += T) {
        const uint32_t key = t.lkey[i];
        if (key == kEmpty) continue;
        const uint32_t b = min(key >> bshift, kSortBuckets - 1);
        uint32_t r = t.bcnt[b];
        for (uint32_t j = t.bhead[b]; j != kEmpty; j = t.lnext[j]) r += (t.lkey[j] < key) ? 1u : 0u;
        const uint32_t slot = key >> 1;
        uint32_t lo = 0, hi = n;
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (t.rowptr[mid] <= slot) lo = mid; else hi = mid;
        }
        const uint32_t o = e_run + r;
        if (o < cape) {
          g_row[o] = lo;
          g_col[o] = t.lval[i];
          g_eid[o] = (key & 1u) ? t.rowe0[lo] + (slot - t.rowptr[lo]) : 0xFFFFFFFFu;   // .cpp:410, :422
        }
      }
    }
    e_run += ctrl[C_MV];
    rq0 = rq1;
    __syncthreads();
    if (rq0 >= Q) break;
  }
  SHD_STAMP(3);   // match sort + write-out done
#ifdef SHADOW_SG_TIMING
  if (threadIdx.x == 0) { res[R_T0 + 4] = tacc[0]; res[R_T0 + 5] = tacc[1]; res[R_T0 + 6] = tacc[2]; res[R_T0 + 7] = tacc[3]; }
#endif
  if (tid == 0) {
    res[R_N] = n; res[R_E] = e_run; res[R_FLAGS] = (e_run > cape) ? 2u : 0u; res[R_SLOTS] = S;
    res[R_FNODES] = ctrl[C_FRONT_NODES]; res[R_FREADS] = ctrl[C_FRONT_READS];
  }
  __syncthreads();
}

// LDS carve shared by host (size computation) and device
struct LdsLayout {
  size_t hkey, hval, pprv, nodes, rowptr, qptr, rowe0, front0, front1, lkey, lval, lnext, bhead,
      bcnt, bits, wtmp, ctrl, wsum, total;
};

__host__ __device__ inline size_t r16(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline LdsLayout lds_layout(uint32_t H, uint32_t capn, uint32_t capf, uint32_t capm,
                                                bool ppr) {
  LdsLayout L;
  size_t o = 0;
  L.hkey = o; o += r16((size_t)(H + kStash) * 4);
  L.hval = o; o += r16((size_t)(H + kStash) * 4);
  L.pprv = o; o += ppr ? r16((size_t)(H + kStash) * 4) : 0;
  L.nodes = o; o += r16((size_t)capn * 4);
  L.rowptr = o; o += r16((size_t)(capn + 1) * 4);
  L.qptr = o; o += r16((size_t)(capn + 1) * 4);
  L.rowe0 = o; o += r16((size_t)capn * 4);
  L.front0 = o;                                     // the frontiers are dead once the node set is
  L.front1 = o + r16((size_t)capf * 4);             // final: the scan's filter reuses their space
  L.bits = o;
  o += (2 * r16((size_t)capf * 4) > (size_t)kBitWords * 4) ? 2 * r16((size_t)capf * 4) : (size_t)kBitWords * 4;
  L.lkey = o; o += r16((size_t)capm * 4);
  L.lval = o; o += r16((size_t)capm * 4);
  L.lnext = o; o += r16((size_t)capm * 4);
  L.bhead = o; o += kSortBuckets * 4;
  L.bcnt = o; o += kSortBuckets * 4;
  L.wtmp = o; o += 16 * 128;
  L.ctrl = o; o += C_WORDS * 4;
  L.wsum = o; o += 32 * 4;
  L.total = o;
  return L;
}

template <bool kPlain>
__global__ void sg_sample_lds_kernel(SampleParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_next;
  const LdsLayout L = lds_layout(p.H, p.capn, p.capf, p.capm, p.method == SG_METHOD_PPR);
  Tables t;
  t.hkey = (uint32_t *)(smem + L.hkey);
  t.hval = (uint32_t *)(smem + L.hval);
  t.pprv = (float *)(smem + L.pprv);
  t.nodes = (uint32_t *)(smem + L.nodes);
  t.rowptr = (uint32_t *)(smem + L.rowptr);
  t.qptr = (uint32_t *)(smem + L.qptr);
  t.rowe0 = (uint32_t *)(smem + L.rowe0);
  t.front0 = (uint32_t *)(smem + L.front0);
  t.front1 = (uint32_t *)(smem + L.front1);
  t.lkey = (uint32_t *)(smem + L.lkey);
  t.lval = (uint32_t *)(smem + L.lval);
  t.lnext = (uint32_t *)(smem + L.lnext);
  t.bhead = (uint32_t *)(smem + L.bhead);
  t.bcnt = (uint32_t *)(smem + L.bcnt);
  t.bits = (uint32_t *)(smem + L.bits);
  t.wtmp = smem + L.wtmp;
  uint32_t *ctrl = (uint32_t *)(smem + L.ctrl);
  uint32_t *wsum = (uint32_t *)(smem + L.wsum);
  // persistent workgroups: subgraph ids come from a global ticket
  for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(p.g_ticket, 1u);
    __syncthreads();
    const uint32_t s = s_next;
    __syncthreads();
    if (s >= p.P) return;
    sample_subgraph<false, kPlain>(p, s, t, ctrl, wsum);
  }
}

// Big path: persistent workgroups pull overflowed subgraphs (flag bit0 from the
// LDS kernel) from a ticket counter and redo them over global-memory tables.
__global__ void sg_sample_big_kernel(SampleParams p) {
  __shared__ uint32_t ctrl[C_WORDS];
  __shared__ uint32_t wsum[32];
  __shared__ uint32_t bhead[kSortBuckets];
  __shared__ uint32_t bcnt[kSortBuckets];
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];     // [kBitWordsBig] words
  uint32_t *bits = reinterpret_cast<uint32_t *>(big_smem);
  __shared__ __attribute__((aligned(16))) unsigned char wtmp[16 * 128];
  __shared__ uint32_t s_next;
  uint32_t *base = p.g_tables + (size_t)blockIdx.x * p.g_stride;
  Tables t;
  size_t o = 0;
  const size_t hs = (size_t)p.H + kStash;
  t.hkey = base + o; o += hs;
  t.hval = base + o; o += hs;
  t.pprv = (float *)(base + o); o += hs;
  t.nodes = base + o; o += (p.capn + 3u) & ~3u;
  t.rowptr = base + o; o += ((size_t)p.capn + 4) & ~(size_t)3;
  t.qptr = base + o; o += ((size_t)p.capn + 4) & ~(size_t)3;
  t.rowe0 = base + o; o += (p.capn + 3u) & ~3u;
  t.front0 = base + o; o += (p.capf + 3u) & ~3u;
  t.front1 = base + o; o += (p.capf + 3u) & ~3u;
  t.lkey = base + o; o += p.capm;
  t.lval = base + o; o += p.capm;
  t.lnext = base + o; o += p.capm;
  t.bhead = bhead;
  t.bcnt = bcnt;
  t.bits = bits;
  t.wtmp = wtmp;
  for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(p.g_ticket, 1u);
    __syncthreads();
    const uint32_t s = s_next;
    __syncthreads();
    if (s >= p.P) return;
    const uint32_t flags = p.s_cnt[(size_t)s * R_WORDS + R_FLAGS];
    if (!(flags & 1u)) continue;
    sample_subgraph<true, false>(p, s, t, ctrl, wsum);
  }
}

}  // namespace shadow
