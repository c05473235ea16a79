// Device-side subgraph cache: the record -> reuse behaviour of the reference's minibatch loop
// for deterministic samplers (CachedSubgraph / PoolSubgraph, shaDow/minibatch.py:21-91,
// par_graph_sample :403-426, collate :42-66 + frontend/graph.py:280-330) kept entirely in HBM.
//
//   epoch 1 ("record")  sg_cache_record appends the subgraphs of a finished block-diagonal batch whose root is not on
//                       file yet to the arena and files them under their root id (id_root = node[target], :411)
//   epoch >= 2 ("reuse") sg_cache_collate rebuilds the block-diagonal batch of any list of roots
//                       from the arena -- no sampling, no full graph needed
//
// Arena layout (uint32 unless noted), all append-only:
//   a_node[Nn]  original node id          a_ptr[Nn]  row start LOCAL to the subgraph
//   a_hop[Nn]   hop (0xFFFFFFFF if the recorded batch had none)   a_ppr[Nn] float
//   a_col[Ne]   column id LOCAL to the subgraph                   a_eid[Ne] original edge id
// Root table (dense over the graph's node ids): t_nstart / t_estart (uint64 arena offsets),
//   t_n / t_e (sizes; t_n == 0xFFFFFFFF: not recorded), t_tgt (local id of the root).
// Both kernels are pure streaming copies with an offset fix-up: HBM-bound.
#include <string.h>

#include <algorithm>

#include "common.h"

using namespace shadow;

struct sg_cache {
  int device = 0;
  uint32_t N = 0;
  uint32_t *a_node = nullptr, *a_ptr = nullptr, *a_hop = nullptr, *a_col = nullptr, *a_eid = nullptr;
  float *a_ppr = nullptr;
  uint64_t cap_n = 0, cap_e = 0, fill_n = 0, fill_e = 0;
  uint64_t *t_nstart = nullptr, *t_estart = nullptr;
  uint32_t *t_n = nullptr, *t_e = nullptr, *t_tgt = nullptr;
  uint64_t num_recorded = 0;
  uint64_t *d_counts = nullptr;     // [8]
  uint64_t *h_counts = nullptr;     // pinned
  hipEvent_t ev = nullptr;
  bool pending = false;
  uint32_t pending_P = 0;
  // record plan: arena offsets of the batch's NEW subgraphs (roots already in the table are skipped)
  uint64_t *d_plan = nullptr;       // [2 * cap_plan]: node / edge destination per subgraph, ~0 = skip
  uint32_t cap_plan = 0;
};

namespace {

constexpr uint32_t kAbsent = 0xFFFFFFFFu;
constexpr uint32_t kCB = 256;

__global__ void cache_init_table_kernel(uint32_t *t_n, uint32_t N) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (uint64_t)gridDim.x * blockDim.x)
    t_n[i] = kAbsent;
}

// one workgroup per subgraph of the batch being recorded
__global__ void __launch_bounds__(kCB)
cache_record_kernel(sg_batch_out b, uint32_t P, const uint64_t *__restrict__ plan, uint32_t *a_node, uint32_t *a_ptr,
                    uint32_t *a_hop, float *a_ppr, uint32_t *a_col, uint32_t *a_eid, uint64_t *t_nstart,
                    uint64_t *t_estart, uint32_t *t_n, uint32_t *t_e, uint32_t *t_tgt, uint32_t N) {
  const uint32_t s = blockIdx.x;
  if (s >= P) return;
  const uint64_t dn = plan[2 * s], de = plan[2 * s + 1];
  if (dn == ~0ull) return;                                   // this root is already on file
  const uint32_t a = b.d_subg_nodes[s], ns = b.d_subg_nodes[s + 1] - a;
  const uint32_t e0 = b.d_subg_edges[s], es = b.d_subg_edges[s + 1] - e0;
  for (uint32_t i = threadIdx.x; i < ns; i += kCB) {
    a_node[dn + i] = b.d_node[a + i];
    a_ptr[dn + i] = b.d_indptr[a + i] - e0;
    a_hop[dn + i] = b.d_hop ? b.d_hop[a + i] : 0xFFFFFFFFu;
    a_ppr[dn + i] = b.d_ppr ? b.d_ppr[a + i] : -1.0f;
  }
  for (uint32_t p = threadIdx.x; p < es; p += kCB) {
    a_col[de + p] = b.d_indices[e0 + p] - a;
    a_eid[de + p] = b.d_edge_id[e0 + p];
  }
  if (threadIdx.x == 0) {
    const uint32_t tg = b.d_target[s];                       // single-root subgraphs (minibatch.py:410)
    const uint32_t root = b.d_node[tg];
    t_nstart[root] = dn; t_estart[root] = de;
    t_e[root] = es; t_tgt[root] = tg - a;
    t_n[root] = ns;
  }
}

// Which subgraphs of the batch are new (their root is not in the table yet -- with percent_per_epoch < 1 most roots of a
// later "record" epoch already are, and appending them again only grew the arena: ADVICE r2) and where they go: exclusive
// prefix of the NEW subgraphs' sizes behind the arena's fill marks.  One workgroup; counts[0..2] = new nodes / edges /
// subgraphs.  (Two subgraphs of one batch with the same new root are both stored; the table keeps the later one.)
__global__ void __launch_bounds__(1024)
cache_record_plan_kernel(sg_batch_out b, uint32_t P, const uint32_t *__restrict__ t_n, uint32_t N, uint64_t fill_n, uint64_t fill_e,
                         uint64_t *__restrict__ plan, uint64_t *__restrict__ counts) {
  __shared__ uint32_t wsum_n[16], wsum_e[16], wsum_c[16];
  __shared__ uint64_t carry_n, carry_e;
  __shared__ uint32_t carry_c;
  const uint32_t tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  if (tid == 0) { carry_n = 0; carry_e = 0; carry_c = 0; }
  __syncthreads();
  for (uint32_t base = 0; base < P; base += 1024) {
    const uint32_t s = base + tid;
    uint32_t n = 0, e = 0, isnew = 0;
    if (s < P) {
      const uint32_t root = b.d_node[b.d_target[s]];
      if (root < N && t_n[root] == kAbsent) {
        isnew = 1; n = b.d_subg_nodes[s + 1] - b.d_subg_nodes[s]; e = b.d_subg_edges[s + 1] - b.d_subg_edges[s];
      }
    }
    const uint32_t in = wave_incl_scan(n), ie = wave_incl_scan(e), ic = wave_incl_scan(isnew);
    if (lane == 63) { wsum_n[wv] = in; wsum_e[wv] = ie; wsum_c[wv] = ic; }
    __syncthreads();
    uint64_t pn = carry_n, pe = carry_e;
    uint32_t pc = carry_c;
    for (uint32_t w = 0; w < wv; w++) { pn += wsum_n[w]; pe += wsum_e[w]; pc += wsum_c[w]; }
    if (s < P) {
      plan[2 * s] = isnew ? fill_n + pn + in - n : ~0ull;
      plan[2 * s + 1] = isnew ? fill_e + pe + ie - e : ~0ull;
    }
    __syncthreads();
    if (tid == 1023) { carry_n = pn + in; carry_e = pe + ie; carry_c = pc + ic; }
    __syncthreads();
  }
  if (tid == 0) { counts[0] = carry_n; counts[1] = carry_e; counts[2] = carry_c; }
}

// sizes of the requested subgraphs + offsets (single workgroup scan) + batch totals
__global__ void __launch_bounds__(1024)
cache_offsets_kernel(const uint32_t *__restrict__ roots, uint32_t P, const uint32_t *__restrict__ t_n,
                     const uint32_t *__restrict__ t_e, uint32_t N, sg_batch_out out, uint64_t *counts) {
  __shared__ uint32_t wsum_n[16], wsum_e[16];
  __shared__ uint32_t carry_n, carry_e, mx_n, mx_e, miss;
  const uint32_t tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  if (tid == 0) { carry_n = 0; carry_e = 0; mx_n = 0; mx_e = 0; miss = 0; }
  __syncthreads();
  for (uint32_t base = 0; base < P; base += 1024) {
    const uint32_t s = base + tid;
    uint32_t n = 0, e = 0;
    if (s < P) {
      const uint32_t r = roots[s];
      const uint32_t tn = r < N ? t_n[r] : kAbsent;
      if (tn == kAbsent) atomicAdd(&miss, 1u);
      else { n = tn; e = t_e[r]; }
      atomicMax(&mx_n, n); atomicMax(&mx_e, e);
    }
    const uint32_t in = wave_incl_scan(n), ie = wave_incl_scan(e);
    if (lane == 63) { wsum_n[wv] = in; wsum_e[wv] = ie; }
    __syncthreads();
    uint32_t pn = carry_n, pe = carry_e;
    for (uint32_t w = 0; w < wv; w++) { pn += wsum_n[w]; pe += wsum_e[w]; }
    if (s < P) { out.d_subg_nodes[s] = pn + in - n; out.d_subg_edges[s] = pe + ie - e; }
    __syncthreads();
    if (tid == 1023) { carry_n = pn + in; carry_e = pe + ie; }
    __syncthreads();
  }
  if (tid == 0) {
    out.d_subg_nodes[P] = carry_n; out.d_subg_edges[P] = carry_e;
    counts[0] = carry_n; counts[1] = carry_e; counts[2] = mx_n; counts[3] = mx_e;
    uint32_t ovf = 0;
    if ((uint64_t)carry_n > out.cap_nodes) ovf |= 4u;
    if ((uint64_t)carry_e > out.cap_edges) ovf |= 8u;
    counts[4] = ovf; counts[5] = miss;
  }
}

// one workgroup per requested subgraph: copy with the offset fix-up of cat_to_block_diagonal
__global__ void __launch_bounds__(kCB)
cache_collate_kernel(const uint32_t *__restrict__ roots, uint32_t P, const uint32_t *a_node, const uint32_t *a_ptr,
                     const uint32_t *a_hop, const float *a_ppr, const uint32_t *a_col, const uint32_t *a_eid,
                     const uint64_t *t_nstart, const uint64_t *t_estart, const uint32_t *t_n, const uint32_t *t_e,
                     const uint32_t *t_tgt, uint32_t N, sg_batch_out out, const uint64_t *counts) {
  if (counts[4] != 0) return;                                  // output too small: nothing is written
  const uint32_t s = blockIdx.x;
  const uint32_t r = roots[s];
  const uint32_t no = out.d_subg_nodes[s], eo = out.d_subg_edges[s];
  if (s == P - 1 && threadIdx.x == 0) out.d_indptr[out.d_subg_nodes[P]] = out.d_subg_edges[P];
  if (r >= N || t_n[r] == kAbsent) { if (threadIdx.x == 0) out.d_target[s] = no; return; }
  const uint32_t ns = t_n[r], es = t_e[r];
  const uint64_t nb = t_nstart[r], eb = t_estart[r];
  for (uint32_t i = threadIdx.x; i < ns; i += kCB) {
    out.d_node[no + i] = a_node[nb + i];
    out.d_indptr[no + i] = a_ptr[nb + i] + eo;
    if (out.d_hop) out.d_hop[no + i] = a_hop[nb + i];
    if (out.d_ppr) out.d_ppr[no + i] = a_ppr[nb + i];
  }
  for (uint32_t p = threadIdx.x; p < es; p += kCB) {
    out.d_indices[eo + p] = a_col[eb + p] + no;
    out.d_edge_id[eo + p] = a_eid[eb + p];
  }
  if (threadIdx.x == 0) out.d_target[s] = no + t_tgt[r];
}

template <typename T>
int grow(T **ptr, uint64_t have_elems, uint64_t want_elems, hipStream_t st) {
  T *n = nullptr;
  SHD_HIP(hipMalloc((void **)&n, want_elems * sizeof(T)));
  if (*ptr && have_elems) SHD_HIP(hipMemcpyAsync(n, *ptr, have_elems * sizeof(T), hipMemcpyDeviceToDevice, st));
  SHD_HIP(hipStreamSynchronize(st));
  if (*ptr) (void)hipFree(*ptr);
  *ptr = n;
  return SG_OK;
}

}  // namespace

extern "C" int sg_cache_create(uint32_t num_nodes, int device_id, sg_cache **out) {
  if (!out) return set_error(SG_ERR_INVALID, "sg_cache_create: null argument");
  *out = nullptr;
  SHD_HIP(hipSetDevice(device_id));
  sg_cache *c = new sg_cache();
  c->device = device_id; c->N = num_nodes;
  const size_t n1 = std::max<size_t>(num_nodes, 1);
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = hipMalloc((void **)&c->t_nstart, n1 * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&c->t_estart, n1 * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&c->t_n, n1 * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&c->t_e, n1 * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&c->t_tgt, n1 * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_counts, 8 * 8);
  if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_counts, 8 * 8, hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    const int rc = set_error(SG_ERR_HIP, "sg_cache_create: %s", hipGetErrorString(e));
    sg_cache_destroy(c);
    return rc;
  }
  hipLaunchKernelGGL(cache_init_table_kernel, dim3(std::min<uint32_t>(4096, (uint32_t)((n1 + 255) / 256))), dim3(256), 0, 0,
                     c->t_n, num_nodes);
  SHD_HIP(hipGetLastError());
  SHD_HIP(hipDeviceSynchronize());
  *out = c;
  return SG_OK;
}

extern "C" void sg_cache_destroy(sg_cache *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  void *ptrs[] = {c->a_node, c->a_ptr, c->a_hop, c->a_col, c->a_eid, c->a_ppr, c->t_nstart, c->t_estart,
                  c->t_n, c->t_e, c->t_tgt, c->d_counts, c->d_plan};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  if (c->h_counts) (void)hipHostFree(c->h_counts);
  if (c->ev) (void)hipEventDestroy(c->ev);
  delete c;
}

extern "C" int sg_cache_clear(sg_cache *c) {
  if (!c) return set_error(SG_ERR_INVALID, "sg_cache_clear: null cache");
  SHD_HIP(hipSetDevice(c->device));
  SHD_HIP(hipDeviceSynchronize());
  c->fill_n = c->fill_e = 0; c->num_recorded = 0;
  hipLaunchKernelGGL(cache_init_table_kernel, dim3(std::min<uint32_t>(4096, (c->N + 255) / 256 + 1)), dim3(256), 0, 0,
                     c->t_n, c->N);
  SHD_HIP(hipGetLastError());
  SHD_HIP(hipDeviceSynchronize());
  return SG_OK;
}

extern "C" int sg_cache_stats(const sg_cache *c, uint64_t *num_recorded, uint64_t *nodes, uint64_t *edges) {
  if (!c) return set_error(SG_ERR_INVALID, "sg_cache_stats: null cache");
  if (num_recorded) *num_recorded = c->num_recorded;
  if (nodes) *nodes = c->fill_n;
  if (edges) *edges = c->fill_e;
  return SG_OK;
}

extern "C" int sg_cache_record(sg_cache *c, const sg_batch_out *batch, uint32_t num_subg, uint64_t n_tot,
                               uint64_t e_tot, void *stream_) {
  if (!c || !batch) return set_error(SG_ERR_INVALID, "sg_cache_record: null argument");
  if (!batch->d_node || !batch->d_indptr || !batch->d_indices || !batch->d_edge_id || !batch->d_target ||
      !batch->d_subg_nodes || !batch->d_subg_edges)
    return set_error(SG_ERR_INVALID, "sg_cache_record: incomplete batch");
  if (num_subg == 0) return SG_OK;
  SHD_HIP(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream_;
  if (c->fill_n + n_tot > c->cap_n) {
    const uint64_t want = std::max<uint64_t>((c->fill_n + n_tot) * 2, 1 << 16);
    int rc;
    if ((rc = grow(&c->a_node, c->fill_n, want, st)) != SG_OK) return rc;
    if ((rc = grow(&c->a_ptr, c->fill_n, want, st)) != SG_OK) return rc;
    if ((rc = grow(&c->a_hop, c->fill_n, want, st)) != SG_OK) return rc;
    if ((rc = grow(&c->a_ppr, c->fill_n, want, st)) != SG_OK) return rc;
    c->cap_n = want;
  }
  if (c->fill_e + e_tot > c->cap_e) {
    const uint64_t want = std::max<uint64_t>((c->fill_e + e_tot) * 2, 1 << 16);
    int rc;
    if ((rc = grow(&c->a_col, c->fill_e, want, st)) != SG_OK) return rc;
    if ((rc = grow(&c->a_eid, c->fill_e, want, st)) != SG_OK) return rc;
    c->cap_e = want;
  }
  if (num_subg > c->cap_plan) {
    if (c->d_plan) (void)hipFree(c->d_plan);
    c->d_plan = nullptr; c->cap_plan = 0;
    const uint32_t want = std::max<uint32_t>(num_subg * 2, 1024);
    SHD_HIP(hipMalloc((void **)&c->d_plan, (size_t)want * 2 * 8));
    c->cap_plan = want;
  }
  if (c->pending) return set_error(SG_ERR_STATE, "sg_cache_record: a collate is in flight (its counters are in use)");
  // only the subgraphs whose root is not on file yet are appended (the capacity above covers the whole batch)
  hipLaunchKernelGGL(cache_record_plan_kernel, dim3(1), dim3(1024), 0, st, *batch, num_subg, c->t_n, c->N, c->fill_n, c->fill_e,
                     c->d_plan, c->d_counts);
  hipLaunchKernelGGL(cache_record_kernel, dim3(num_subg), dim3(kCB), 0, st, *batch, num_subg, c->d_plan, c->a_node, c->a_ptr,
                     c->a_hop, c->a_ppr, c->a_col, c->a_eid, c->t_nstart, c->t_estart, c->t_n, c->t_e, c->t_tgt, c->N);
  SHD_HIP(hipGetLastError());
  SHD_HIP(hipMemcpyAsync(c->h_counts, c->d_counts, 3 * 8, hipMemcpyDeviceToHost, st));
  SHD_HIP(hipStreamSynchronize(st));               // (record epochs only; the sampler call before it has synchronised anyway)
  c->fill_n += c->h_counts[0]; c->fill_e += c->h_counts[1]; c->num_recorded += c->h_counts[2];
  return SG_OK;
}

extern "C" int sg_cache_collate(sg_cache *c, const uint32_t *d_roots, uint32_t num_subg, sg_batch_out *out,
                                void *stream_) {
  if (!c || !d_roots || !out) return set_error(SG_ERR_INVALID, "sg_cache_collate: null argument");
  if (!out->d_node || !out->d_indptr || !out->d_indices || !out->d_edge_id || !out->d_target || !out->d_subg_nodes ||
      !out->d_subg_edges)
    return set_error(SG_ERR_INVALID, "sg_cache_collate: incomplete output descriptor");
  if (c->pending) return set_error(SG_ERR_STATE, "sg_cache_collate: a collate is already in flight");
  if (num_subg == 0) return set_error(SG_ERR_INVALID, "sg_cache_collate: no roots");
  SHD_HIP(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream_;
  hipLaunchKernelGGL(cache_offsets_kernel, dim3(1), dim3(1024), 0, st, d_roots, num_subg, c->t_n, c->t_e, c->N, *out,
                     c->d_counts);
  SHD_HIP(hipGetLastError());
  hipLaunchKernelGGL(cache_collate_kernel, dim3(num_subg), dim3(kCB), 0, st, d_roots, num_subg, c->a_node, c->a_ptr,
                     c->a_hop, c->a_ppr, c->a_col, c->a_eid, c->t_nstart, c->t_estart, c->t_n, c->t_e, c->t_tgt, c->N,
                     *out, c->d_counts);
  SHD_HIP(hipGetLastError());
  SHD_HIP(hipMemcpyAsync(c->h_counts, c->d_counts, 8 * 8, hipMemcpyDeviceToHost, st));
  SHD_HIP(hipEventRecord(c->ev, st));
  c->pending = true; c->pending_P = num_subg;
  return SG_OK;
}

extern "C" int sg_cache_collate_finish(sg_cache *c, sg_batch_counts *counts) {
  if (!c || !counts) return set_error(SG_ERR_INVALID, "sg_cache_collate_finish: null argument");
  if (!c->pending) return set_error(SG_ERR_STATE, "sg_cache_collate_finish: nothing in flight");
  SHD_HIP(hipSetDevice(c->device));
  SHD_HIP(hipEventSynchronize(c->ev));
  c->pending = false;
  const uint64_t *h = c->h_counts;
  memset(counts, 0, sizeof(*counts));
  counts->n_tot = h[0]; counts->e_tot = h[1];
  counts->num_subgraphs = c->pending_P;
  counts->max_subg_nodes = (uint32_t)h[2]; counts->max_subg_edges = (uint32_t)h[3];
  counts->overflow = (uint32_t)h[4];
  if (h[5] != 0)
    return set_error(SG_ERR_STATE, "sg_cache_collate: %llu of %u roots were never recorded", (unsigned long long)h[5],
                     c->pending_P);
  if (counts->overflow)
    return set_error(SG_ERR_CAPACITY, "sg_cache_collate: output too small (flags=0x%x) for %llu nodes / %llu edges",
                     counts->overflow, (unsigned long long)counts->n_tot, (unsigned long long)counts->e_tot);
  return SG_OK;
}
