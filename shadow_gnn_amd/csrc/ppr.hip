// ppr.hip -- approximate personalised PageRank by lazy-walk push on the MI355X,
// behind sg_ppr_push.
//
// Replaces ParallelSampler::preproc_ppr_approximate
// (para_graph_sampler/graph_engine/backend/ParallelSampler.cpp:237-340): the
// reference runs one OpenMP thread per target with dense per-thread vectors of
// size N and a std::set as the pending structure.  Here ONE WAVEFRONT owns one
// target ("ordered" mode: bit-exact with the reference):
//   * the sparse state (pi, residue, flags) lives in a per-wave open-addressing
//     hash table in HBM -- no O(N) allocation per target;
//   * the node pushed next is always the smallest id in the pending set
//     (std::set::begin, .cpp:271-273): a binary min-heap with lazy deletion,
//     operated by lane 0;
//   * the neighbour loop of a push (.cpp:287-304) is spread over the 64 lanes:
//     coalesced row read, one hash probe and one fp32 add per neighbour;
//   * every fp32 operation is issued with explicit round-to-nearest intrinsics
//     in the reference's order (no FMA contraction), so scores are bit-identical.
// "fifo" mode (sg_ppr_push mode 1) keeps everything except the ORDER of the pushes: the pending set is a
// ring queue in discovery order, filled by all lanes at once (ballot + prefix count) instead of a heap that
// lane 0 maintains.  Any push order ends with every residue <= epsilon * degree (Andersen-Chung-Lang), so the
// scores agree with the ordered mode within the approximation error, not bit for bit: this mode is validated
// by tolerance (tests/test_sampler_gpu.py) and is several times faster (no O(log n) chain of dependent HBM
// accesses per push).
// The touched set (node, pi) of every target is appended to a flat output list;
// the top-k ordering (-score, id) is done by the caller.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.h"

// The scores must round exactly like the reference's scalar fp32 code: no fused
// multiply-add anywhere in this file (hipcc contracts a*b+c by default; HIP's __fmul_rn /
// __fadd_rn are header inlines that still carry the contract flag, so plain operators under
// this pragma are used instead).
#pragma clang fp contract(off)

namespace shadow {

constexpr uint32_t kPprEmpty = 0xFFFFFFFFu;

struct PprParams {
  const uint32_t *indptr, *indices;
  uint32_t N;
  const uint32_t *targets;
  uint32_t T;
  float alpha1;          // 1 - alpha (the reference flips alpha on entry, .cpp:242)
  float epsilon;
  // per-wave state
  uint32_t H;            // hash slots per wave (power of two)
  uint32_t hshift;
  uint32_t *keys;        // [W*H]
  float *pi, *res;       // [W*H]
  uint8_t *flags;        // [W*H] bit0 in pending set, bit1 pushed at least once, bit2 physically in the heap
  uint32_t *used;        // [W*H] list of occupied slots (for the reset)
  uint32_t *heap;        // [W*H]
  // outputs
  uint32_t *out_count;   // [T] touched nodes of each target
  uint64_t *out_offset;  // [T] position of the target's entries in out_node/out_score
  uint32_t *out_node;    // [cap_out]
  float *out_score;      // [cap_out]
  uint64_t cap_out;
  unsigned long long *cursor;   // [0] output cursor, [1] overflow flags
  uint32_t *ticket;
};

__device__ __forceinline__ uint32_t ppr_hash(uint32_t k, uint32_t hshift) { return (k * 0x9E3779B1u) >> hshift; }

// find or insert `key`; new entries start with pi = res = 0.  Returns the slot or
// kPprEmpty when the table is full.
__device__ __forceinline__ uint32_t ppr_slot(const PprParams &p, uint32_t *keys, float *pi, float *res,
                                             uint8_t *flags, uint32_t *used, uint32_t *n_used,
                                             uint32_t key, bool insert) {
  const uint32_t mask = p.H - 1;
  uint32_t s = ppr_hash(key, p.hshift);
  for (uint32_t probe = 0; probe < p.H; probe++, s = (s + 1) & mask) {
    // keys change through L2 atomics: read them past the L1 (agent scope), never a stale line
    uint32_t k = __hip_atomic_load(&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) return s;
    if (k == kPprEmpty) {
      if (!insert) return kPprEmpty;
      const uint32_t old = atomicCAS(&keys[s], kPprEmpty, key);
      if (old == kPprEmpty) {
        pi[s] = 0.f; res[s] = 0.f; flags[s] = 0;
        const uint32_t i = atomicAdd(n_used, 1u);
        used[i] = s;
        return s;
      }
      if (old == key) return s;
    }
  }
  return kPprEmpty;
}

__device__ __forceinline__ void heap_push(uint32_t *h, uint32_t *n, uint32_t x) {
  uint32_t i = (*n)++;
  h[i] = x;
  while (i > 0) {
    const uint32_t par = (i - 1) >> 1;
    const uint32_t a = h[par], b = h[i];
    if (a <= b) break;
    h[par] = b; h[i] = a; i = par;
  }
}

__device__ __forceinline__ void heap_pop(uint32_t *h, uint32_t *n) {
  const uint32_t m = --(*n);
  h[0] = h[m];
  uint32_t i = 0;
  for (;;) {
    const uint32_t l = 2 * i + 1, r = l + 1;
    uint32_t s = i;
    if (l < m && h[l] < h[s]) s = l;
    if (r < m && h[r] < h[s]) s = r;
    if (s == i) break;
    const uint32_t t = h[s]; h[s] = h[i]; h[i] = t; i = s;
  }
}

template <bool kFifo>
__global__ void ppr_push_kernel(PprParams p) {
  __shared__ uint32_t s_nused[16], s_heapn[16], s_cur[16], s_fail[16], s_head[16];
  const uint32_t lane = lane_id(), wv = wave_id();
  const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + wv;
  uint32_t *keys = p.keys + (size_t)gw * p.H;
  float *pi = p.pi + (size_t)gw * p.H, *res = p.res + (size_t)gw * p.H;
  uint8_t *flags = p.flags + (size_t)gw * p.H;
  uint32_t *used = p.used + (size_t)gw * p.H, *heap = p.heap + (size_t)gw * p.H;
  uint32_t *n_used = &s_nused[wv], *heap_n = &s_heapn[wv];
  for (;;) {
    uint32_t ti = 0;
    if (lane == 0) ti = atomicAdd(p.ticket, 1u);
    ti = __builtin_amdgcn_readfirstlane(ti);
    if (ti >= p.T) return;
    const uint32_t target = p.targets[ti];
    if (lane == 0) {
      *n_used = 0; *heap_n = 0; s_fail[wv] = 0; s_head[wv] = 0;
      const uint32_t st = ppr_slot(p, keys, pi, res, flags, used, n_used, target, true);
      res[st] = 1.0f;                                              // .cpp:269
      flags[st] = 1 | 4;
      if (kFifo) { heap[0] = target; *heap_n = 1; }                // ring queue: [s_head, heap_n) mod H
      else heap_push(heap, heap_n, target);                        // .cpp:271
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (uint32_t iter = 0;; iter++) {
      if (iter > (1u << 26)) { if (lane == 0) s_fail[wv] = 1; break; }     // safety cap: never spin forever
      // smallest pending id (lazy deletion: drop entries that left the set)
      if (kFifo) {
        if (lane == 0) {
          uint32_t cur = kPprEmpty;
          if (s_head[wv] != *heap_n) { cur = heap[s_head[wv] & (p.H - 1)]; s_head[wv]++; }
          s_cur[wv] = cur;
        }
      } else if (lane == 0) {
        uint32_t cur = kPprEmpty;
        while (*heap_n > 0) {
          const uint32_t v = heap[0];
          const uint32_t sv = ppr_slot(p, keys, pi, res, flags, used, n_used, v, false);
          if (sv == kPprEmpty) { s_fail[wv] = 2; break; }           // cannot happen: every heap entry was inserted
          if (flags[sv] & 1u) { cur = v; break; }
          flags[sv] &= ~4u;                                        // stale entry leaves the heap
          heap_pop(heap, heap_n);
        }
        s_cur[wv] = cur;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      const uint32_t v = s_cur[wv];
      if (v == kPprEmpty || s_fail[wv]) break;                     // .cpp:272
      const uint32_t sv = ppr_slot(p, keys, pi, res, flags, used, n_used, v, false);
      if (sv == kPprEmpty) break;
      const float r0 = res[sv];                                    // .cpp:283
      const uint32_t e0 = p.indptr[v], degv = p.indptr[v + 1] - e0;
      if (lane == 0) {
        pi[sv] = pi[sv] + p.alpha1 * r0;       // .cpp:284
        flags[sv] |= 2u;
      }
      const float m = ((1.0f - p.alpha1) * r0) / (float)(2u * degv);   // .cpp:286
      for (uint32_t base = 0; base < degv; base += 64) {
        const uint32_t e = base + lane;
        bool want = false;
        uint32_t u = 0;
        if (e < degv) {
          u = p.indices[e0 + e];
          const uint32_t su = ppr_slot(p, keys, pi, res, flags, used, n_used, u, true);
          if (su == kPprEmpty) { s_fail[wv] = 1; }
          else {
            const float nr = res[su] + m;                // .cpp:299
            res[su] = nr;
            const uint32_t degu = p.indptr[u + 1] - p.indptr[u];
            const uint32_t fl = flags[su];
            if (nr > p.epsilon * (float)degu && !(fl & 1u)) {                                          // .cpp:300-301
              // re-entering the set: a stale heap entry (bit2) becomes valid again, else push one
              flags[su] = (uint8_t)(fl | 1u | 4u);
              want = !(fl & 4u);
            }
          }
        }
        uint64_t mask = __ballot(want);
        if (kFifo) {
          // every lane appends its own newly pending node (at most H nodes are pending at once: one per slot)
          const uint32_t tail = *heap_n;
          if (want) heap[(tail + (uint32_t)__popcll(mask & lanemask_lt())) & (p.H - 1)] = u;
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) *heap_n = tail + (uint32_t)__popcll(mask);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        } else {
          // lane 0 inserts the newly pending nodes into the heap
          while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            const uint32_t x = __shfl(u, l, 64);
            if (lane == 0) heap_push(heap, heap_n, x);
            mask &= mask - 1;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      if (lane == 0) {
        const float nr = (r0 * (1.0f - p.alpha1)) / 2.0f;   // .cpp:312
        res[sv] = nr;
        if (nr <= p.epsilon * (float)degv) flags[sv] &= ~1u;               // .cpp:313-314 (lazy erase)
        else if (kFifo) { heap[*heap_n & (p.H - 1)] = v; (*heap_n)++; }     // still pending: back of the queue
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    // emit the touched set: every node that was pushed at least once (.cpp:315)
    const uint32_t nu = *n_used;
    uint32_t cnt = 0;
    for (uint32_t i = lane; i < nu; i += 64) cnt += (flags[used[i]] & 2u) ? 1u : 0u;
    cnt = wave_reduce_sum(cnt);
    unsigned long long off = 0;
    if (lane == 0) {
      off = atomicAdd(&p.cursor[0], (unsigned long long)cnt);
      p.out_count[ti] = s_fail[wv] ? 0xFFFFFFFFu : cnt;
      p.out_offset[ti] = off;
      if (s_fail[wv]) atomicOr(&p.cursor[1], 1ull);
      if (off + cnt > p.cap_out) atomicOr(&p.cursor[1], 2ull);
    }
    off = ((unsigned long long)__shfl((uint32_t)(off >> 32), 0, 64) << 32) | __shfl((uint32_t)off, 0, 64);
    uint32_t run = 0;
    for (uint32_t base = 0; base < nu; base += 64) {
      const uint32_t i = base + lane;
      const bool on = i < nu && (flags[used[i]] & 2u);
      const uint64_t mk = __ballot(on);
      if (on) {
        const uint64_t o = off + run + __popcll(mk & lanemask_lt());
        if (o < p.cap_out) {
          p.out_node[o] = __hip_atomic_load(&keys[used[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          p.out_score[o] = pi[used[i]];
        }
      }
      run += __popcll(mk);
    }
    // reset the slots this target used
    for (uint32_t i = lane; i < nu; i += 64) __hip_atomic_store(&keys[used[i]], kPprEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  }
}

__global__ void ppr_init_kernel(uint32_t *keys, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) keys[i] = kPprEmpty;
}

}  // namespace shadow

using namespace shadow;

extern "C" int sg_ppr_push(const uint32_t *d_indptr, const uint32_t *d_indices, uint32_t num_nodes,
                           const uint32_t *d_targets, uint32_t num_targets, float alpha, float epsilon,
                           uint32_t hash_slots, uint32_t num_waves, void *d_work, uint64_t work_bytes,
                           uint32_t *d_out_count, uint64_t *d_out_offset, uint32_t *d_out_node,
                           float *d_out_score, uint64_t cap_out, uint64_t *h_total, uint32_t *h_flags,
                           uint32_t mode, void *stream_) {
  if (!d_indptr || !d_indices || !d_targets || !d_work || !d_out_count || !d_out_offset || !d_out_node ||
      !d_out_score || !h_total || !h_flags)
    return set_error(SG_ERR_INVALID, "sg_ppr_push: null argument");
  if (hash_slots < 64 || (hash_slots & (hash_slots - 1))) return set_error(SG_ERR_INVALID, "sg_ppr_push: hash_slots must be a power of two >= 64");
  if (num_waves == 0 || num_waves % 4) return set_error(SG_ERR_INVALID, "sg_ppr_push: num_waves must be a multiple of 4");
  if (mode > 1) return set_error(SG_ERR_INVALID, "sg_ppr_push: mode %u (0 = ordered / bit-exact, 1 = fifo)", mode);
  hipStream_t st = (hipStream_t)stream_;
  const size_t WH = (size_t)num_waves * hash_slots;
  const size_t need = WH * (4 + 4 + 4 + 4 + 4) + WH + 256;
  if (work_bytes < need) return set_error(SG_ERR_CAPACITY, "sg_ppr_push: work buffer %llu B < %zu B", (unsigned long long)work_bytes, need);
  char *w = (char *)d_work;
  PprParams p;
  memset(&p, 0, sizeof(p));
  p.indptr = d_indptr; p.indices = d_indices; p.N = num_nodes; p.targets = d_targets; p.T = num_targets;
  p.alpha1 = 1 - alpha;                                    // .cpp:242
  p.epsilon = epsilon;
  p.H = hash_slots; p.hshift = 32; for (uint32_t h = hash_slots; h > 1; h >>= 1) p.hshift--;
  p.cursor = (unsigned long long *)w; p.ticket = (uint32_t *)(w + 32);
  size_t o = 256;
  p.keys = (uint32_t *)(w + o); o += WH * 4;
  p.pi = (float *)(w + o); o += WH * 4;
  p.res = (float *)(w + o); o += WH * 4;
  p.used = (uint32_t *)(w + o); o += WH * 4;
  p.heap = (uint32_t *)(w + o); o += WH * 4;
  p.flags = (uint8_t *)(w + o);
  p.out_count = d_out_count; p.out_offset = d_out_offset; p.out_node = d_out_node; p.out_score = d_out_score;
  p.cap_out = cap_out;
  SHD_HIP(hipMemsetAsync(w, 0, 256, st));
  hipLaunchKernelGGL(ppr_init_kernel, dim3(1024), dim3(256), 0, st, p.keys, WH);
  if (mode == 1) hipLaunchKernelGGL(ppr_push_kernel<true>, dim3(num_waves / 4), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(ppr_push_kernel<false>, dim3(num_waves / 4), dim3(256), 0, st, p);
  SHD_HIP(hipGetLastError());
  unsigned long long h[2] = {0, 0};
  SHD_HIP(hipMemcpyAsync(h, p.cursor, 16, hipMemcpyDeviceToHost, st));
  SHD_HIP(hipStreamSynchronize(st));
  *h_total = h[0];
  *h_flags = (uint32_t)h[1];
  if (h[1] & 1ull) return set_error(SG_ERR_CAPACITY, "sg_ppr_push: per-target hash table (%u slots) too small", hash_slots);
  if (h[1] & 2ull) return set_error(SG_ERR_CAPACITY, "sg_ppr_push: output capacity %llu < %llu entries", (unsigned long long)cap_out, h[0]);
  return SG_OK;
}
