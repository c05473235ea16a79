// sampler.hip -- MI355X (gfx950) subgraph sampler behind the sg_* C ABI.
//
// Replaces the reference's C++/OpenMP backend
// (para_graph_sampler/graph_engine/backend/ParallelSampler.cpp) -- it is a new
// design, not a translation:
//
//   * the full-graph CSR lives in HBM; one WORKGROUP samples one subgraph;
//   * node selection (k-hop frontier expansion .cpp:510-547, PPR top-k
//     .cpp:565-590, nodeIID .cpp:498-505) dedupes through an LDS-resident
//     open-addressing hash set; frontiers are LDS lists;
//   * the selected ids are bitonic-sorted in LDS (== std::sort, .cpp:362) and
//     the hash value of each id becomes its rank (orig2subID, .cpp:369-372);
//   * node-induced slicing (.cpp:378-431) is ONE ordered streaming pass over
//     the concatenated full-graph rows of the subgraph's nodes: every stream
//     slot is a coalesced uint32 load from HBM + an LDS hash probe; wavefront
//     ballots + a tiny LDS scan give every surviving edge its output rank, so
//     the emitted CSR is in exactly the reference's order with no atomics;
//   * a second, light kernel relocates the per-subgraph results into the
//     block-diagonal batch layout (frontend/graph.py:280-320), and runs the
//     hop BFS / DRNL labels (Graph.cpp:32-73) on the emitted CSR.
//
// Subgraphs whose node set does not fit the LDS tables are re-sampled by the
// same code instantiated over global-memory tables (sg_sample_big_kernel).
//
// Budgeted k-hop draws come from Philox4x32-10 keyed on
// (seed, subgraph serial, level, node, draw) -- see DESIGN.md "RNG contract".
#include <fcntl.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace shadow {

std::string &last_error() {
  static thread_local std::string e;
  return e;
}

int set_error(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t &have = granted[std::make_pair(kernel, dev)];
  if (bytes <= have) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

}  // namespace shadow

#include "sampler_scan.h"

namespace shadow {

// ---------------------------------------------------------------------------
// Relocation into the block-diagonal batch (frontend/graph.py:280-320), hop
// BFS (Graph.cpp:32-64) and DRNL (Graph.cpp:66-73, .cpp:438-451).
// ---------------------------------------------------------------------------
constexpr uint32_t kMaxBatches = SG_MAX_BATCHES_PER_CALL;

struct RelocParams {
  uint32_t P;
  int R;
  int aug_flags;
  uint32_t cap_nodes_scr, cap_edges_scr;
  const uint32_t *s_nodes;
  const float *s_ppr;
  uint32_t *s_rowptr;     // [P*(cap_nodes_scr+1)] written here (lower_bound over s_row)
  const uint32_t *s_row;
  const uint32_t *s_col;
  const uint32_t *s_eid;
  const uint32_t *s_tgt;
  const uint32_t *s_cnt;
  const uint32_t *cstart;    // [P+1] chunk prefix of the subgraphs
  const RoundRec *recs;      // round records: where each round's ordered survivors sit in the edge scratch
  const uint2 *blkinfo;
  const uint32_t *plan;
  uint32_t *s_tmp;       // [P*cap_nodes_scr] BFS scratch (drnl)
  uint32_t *s_lcol;      // [P*cap_edges_scr] local column ids in final order (hop / drnl BFS)
  // One call may fill SEVERAL batches (sg_sample_multi): the subgraphs [bstart[b], bstart[b+1]) of the call form batch b,
  // written block-diagonally into outs[b] with offsets relative to that batch, counted in d_counts[8 b ...].
  uint32_t nbatch;
  uint32_t bstart[kMaxBatches + 1];
  sg_batch_out outs[kMaxBatches];
  uint64_t *d_counts;    // [nbatch][8] n_tot, e_tot, max_n, max_e, overflow, slots, fnodes, freads
};

__device__ __forceinline__ void bfs_local(const uint32_t *rowptr, const uint32_t *col, uint32_t n,
                                          uint32_t src, uint32_t *hop, uint32_t *changed) {
  const uint32_t tid = threadIdx.x, T = blockDim.x;
  for (uint32_t i = tid; i < n; i += T) hop[i] = (i == src) ? 0u : 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t cur = 0;; cur++) {
    if (tid == 0) *changed = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += T) {
      if (hop[i] == cur) {
        for (uint32_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
          const uint32_t u = col[e];
          if (hop[u] == 0xFFFFFFFFu) { hop[u] = cur + 1; *changed = 1; }
        }
      }
    }
    __syncthreads();
    const uint32_t ch = *changed;
    __syncthreads();
    if (!ch) break;
  }
}

__global__ void sg_relocate_kernel(RelocParams p) {
  __shared__ uint64_t red[4 * 16];
  __shared__ uint64_t stat[16][5];
  __shared__ uint64_t offs[2];
  __shared__ uint32_t changed;
  const uint32_t s = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const uint32_t lane = lane_id(), wave = wave_id(), nw = T >> 6;
  // the batch this subgraph belongs to (uniform; a handful of batches per call)
  uint32_t bi = 0;
  while (bi + 1 < p.nbatch && s >= p.bstart[bi + 1]) bi++;
  const uint32_t s0 = p.bstart[bi], s1 = p.bstart[bi + 1], sb = s - s0, Pb = s1 - s0;
  uint64_t *const counts = p.d_counts + 8 * bi;
  // exclusive prefix of (nodes, edges) over the preceding subgraphs OF THE BATCH; the batch's LAST workgroup also
  // folds the batch statistics of every subgraph (same pass, no contended global atomics)
  const bool last = (s + 1 == s1);
  uint64_t pn = 0, pe = 0, sl = 0, fn = 0, fr = 0;
  uint32_t mxn = 0, mxe = 0;
  for (uint32_t q = s0 + tid; q < s + (last ? 1u : 0u); q += T) {
    const uint32_t *c = p.s_cnt + (size_t)q * R_WORDS;
    if (q < s) { pn += c[R_N]; pe += c[R_E]; }
    if (last) { sl += c[R_SLOTS]; fn += c[R_FNODES]; fr += c[R_FREADS]; mxn = max(mxn, c[R_N]); mxe = max(mxe, c[R_E]); }
  }
  for (int off = 32; off >= 1; off >>= 1) { pn += __shfl_xor(pn, off, 64); pe += __shfl_xor(pe, off, 64); }
  if (lane == 0) { red[wave] = pn; red[16 + wave] = pe; }
  if (last) {
    for (int off = 32; off >= 1; off >>= 1) {
      sl += __shfl_xor(sl, off, 64); fn += __shfl_xor(fn, off, 64); fr += __shfl_xor(fr, off, 64);
      mxn = max(mxn, (uint32_t)__shfl_xor((int)mxn, off, 64)); mxe = max(mxe, (uint32_t)__shfl_xor((int)mxe, off, 64));
    }
    if (lane == 0) { stat[wave][0] = sl; stat[wave][1] = fn; stat[wave][2] = fr; stat[wave][3] = mxn; stat[wave][4] = mxe; }
  }
  __syncthreads();
  if (tid == 0) {
    uint64_t a = 0, b = 0;
    for (uint32_t w = 0; w < nw; w++) { a += red[w]; b += red[16 + w]; }
    offs[0] = a; offs[1] = b;
  }
  __syncthreads();
  const uint64_t noff = offs[0], eoff = offs[1];
  const uint32_t *c = p.s_cnt + (size_t)s * R_WORDS;
  const uint32_t n = c[R_N], e = c[R_E], flags = c[R_FLAGS];
  const sg_batch_out &o = p.outs[bi];
  if (tid == 0) {
    o.d_subg_nodes[sb] = (uint32_t)noff;
    o.d_subg_edges[sb] = (uint32_t)eoff;
    if (last) {
      o.d_subg_nodes[Pb] = (uint32_t)(noff + n); o.d_subg_edges[Pb] = (uint32_t)(eoff + e);
      uint64_t t0 = 0, t1 = 0, t2 = 0, m0 = 0, m1 = 0;
      for (uint32_t w = 0; w < nw; w++) {
        t0 += stat[w][0]; t1 += stat[w][1]; t2 += stat[w][2]; m0 = max(m0, stat[w][3]); m1 = max(m1, stat[w][4]);
      }
      counts[0] = noff + n; counts[1] = eoff + e;
      counts[2] = m0; counts[3] = m1; counts[5] = t0; counts[6] = t1; counts[7] = t2;
    }
  }
  uint32_t ovf = flags & 3u;
  if (e > p.cap_edges_scr) ovf |= 2u;                       // the scan ran out of per-subgraph edge scratch
  if (p.plan[PL_FLAGS] & 16u) ovf |= 16u;                   // ... or of round records
  if (noff + n > o.cap_nodes) ovf |= 4u;
  if (eoff + e > o.cap_edges) ovf |= 8u;
  if (ovf) {
    if (tid == 0) atomicOr((unsigned long long *)&counts[4], (unsigned long long)ovf);
    return;
  }
  const uint32_t *nodes = p.s_nodes + (size_t)s * p.cap_nodes_scr;
  const float *ppr = p.s_ppr + (size_t)s * p.cap_nodes_scr;
  uint32_t *rowptr = p.s_rowptr + (size_t)s * (p.cap_nodes_scr + 1);
  const uint32_t *erow = p.s_row + (size_t)s * p.cap_edges_scr;
  const uint32_t *col = p.s_col + (size_t)s * p.cap_edges_scr;
  const uint32_t *eid = p.s_eid + (size_t)s * p.cap_edges_scr;
  for (uint32_t i = tid; i < n; i += T) {
    o.d_node[noff + i] = nodes[i];
    if (o.d_ppr) o.d_ppr[noff + i] = ppr[i];
  }
  // The scan left the subgraph's edges as round records, every round in the reference's edge order and the rounds
  // ordered by quad position (workgroup by workgroup, in filing order): concatenating them is the ordered edge list.
  // Row pointers come from the same pass: edges are ordered by row, so rowptr[i] = first edge j with row(j) >= i --
  // every edge that starts a new row fills the pointers of the rows since the previous edge's row.
  // hop BFS / DRNL run on the subgraph's own CSR: local column ids in final order
  uint32_t *lcol = p.s_lcol + (size_t)s * p.cap_edges_scr;
  const bool want_lcol = (p.aug_flags & (SG_AUG_HOPS | SG_AUG_DRNLS)) != 0;
  uint32_t dst = 0, prev_row = 0xFFFFFFFFu;                  // uniform walk state
  const uint32_t ch0 = p.cstart[s], ch1 = p.cstart[s + 1], cpw = p.plan[PL_CPW];
  if (ch1 > ch0) {
    for (uint32_t w = ch0 / cpw; w <= (ch1 - 1u) / cpw; w++) {          // the scan workgroups that held chunks of s
      for (uint32_t blk = w; blk != 0xFFFFFFFFu;) {
        const uint2 bi = p.blkinfo[blk];
        for (uint32_t q = 0; q < bi.x; q++) {
          const RoundRec rr = p.recs[(size_t)blk * kRecPerBlock + q];
          if (rr.s != s) continue;
          for (uint32_t k = tid; k < rr.cnt; k += T) {
            const uint32_t src = rr.src_off + k, j = dst + k;
            const uint32_t r_hi = erow[src];
            const uint32_t r_before = (k > 0) ? erow[src - 1] : prev_row;
            const uint32_t r_lo = (j > 0) ? r_before + 1u : 0u;
            for (uint32_t i = r_lo; i <= r_hi && i <= n; i++) rowptr[i] = j;
            const uint32_t cj = col[src];
            o.d_indices[eoff + j] = (uint32_t)(noff + cj);
            o.d_edge_id[eoff + j] = eid[src];
            if (want_lcol) lcol[j] = cj;
          }
          if (rr.cnt) prev_row = erow[rr.src_off + rr.cnt - 1u];
          dst += rr.cnt;
        }
        blk = bi.y;
      }
    }
  }
  for (uint32_t i = (e > 0 ? prev_row + 1u : 0u) + tid; i <= n; i += T) rowptr[i] = e;   // rows behind the last edge
  __syncthreads();
  for (uint32_t i = tid; i < n; i += T) o.d_indptr[noff + i] = (uint32_t)(eoff + rowptr[i]);
  if (last && tid == 0) o.d_indptr[noff + n] = (uint32_t)(eoff + e);
  const uint32_t *tgt = p.s_tgt + (size_t)s * kMaxRoots;
  if (tid < (uint32_t)p.R) o.d_target[(size_t)sb * p.R + tid] = (uint32_t)(noff + tgt[tid]);
  if ((p.aug_flags & SG_AUG_HOPS) && o.d_hop) {
    bfs_local(rowptr, lcol, n, tgt[0], o.d_hop + noff, &changed);           // .cpp:433-436
  }
  if ((p.aug_flags & SG_AUG_DRNLS) && o.d_drnl && p.R >= 2) {               // .cpp:438-451
    uint32_t *dx = p.s_tmp + (size_t)s * p.cap_nodes_scr;
    uint32_t *dy = o.d_drnl + noff;
    bfs_local(rowptr, lcol, n, tgt[0], dx, &changed);
    bfs_local(rowptr, lcol, n, tgt[1], dy, &changed);
    for (uint32_t i = tid; i < n; i += T) {
      const uint32_t a = dx[i], b = dy[i];
      uint32_t r;
      if (a >= 255u || b >= 255u) r = 255u;
      else { const uint32_t d = a + b, mn = a < b ? a : b; r = 1u + mn + (d / 2) * ((d / 2) + (d % 2) - 1u); }
      dy[i] = r;
    }
  }
}

}  // namespace shadow

// ===========================================================================
// Host side
// ===========================================================================
using namespace shadow;

struct sg_sampler {
  int device = 0;
  uint32_t N = 0;
  uint64_t nnz = 0;
  uint32_t *d_indptr = nullptr;
  uint32_t *d_indices = nullptr;
  uint32_t *d_self_slot = nullptr;   // [N] where the reference inserts node v's self edge (first add_self_edge call on the flat scan builds it)
  bool owns_graph = false;
  bool graph_dropped = false;
  uint32_t *d_targets = nullptr;
  uint64_t num_targets = 0, cap_targets = 0;
  uint64_t idx_root = 0;
  uint64_t seed = 0, serial_next = 0;
  // ppr table (device) + host mirror for sg_save_ppr_bin
  int32_t *d_ppr_row = nullptr;
  uint32_t *d_ppr_len = nullptr, *d_ppr_neigh = nullptr;
  float *d_ppr_score = nullptr;
  uint32_t ppr_stride = 0, ppr_rows = 0;
  std::vector<uint32_t> h_ppr_targets, h_ppr_len, h_ppr_neigh;
  std::vector<float> h_ppr_score;
  // scratch
  void *d_scratch = nullptr;
  size_t scratch_bytes = 0;
  void *d_big = nullptr;
  size_t big_bytes = 0;
  uint32_t user_cap_nodes = 0, user_cap_edges = 0;
  uint32_t rec_scale = 1;            // round-record pool multiplier: doubled by sg_sample_finish on overflow flag 16
  uint64_t *d_counts = nullptr;   // [kMaxBatches][8] + tickets
  uint64_t *h_counts = nullptr;   // pinned
  hipEvent_t ev = nullptr;
  bool pending = false;
  uint32_t pending_P = 0;
  uint32_t pending_nbatch = 0;
  uint32_t pending_bsize[kMaxBatches] = {0};
  const uint32_t *last_cnt = nullptr;   // per-subgraph result words of the last call
  const uint32_t *last_plan = nullptr;  // plan words of the last call (item count, phase cycles of the scan)
  bool profiling = false;
  hipEvent_t ev_t[3] = {nullptr, nullptr, nullptr};
  bool timed = false;
};

static uint32_t next_pow2(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return (uint32_t)p;
}

static int ensure(void **ptr, size_t *have, size_t need) {
  if (*have >= need) return SG_OK;
  if (*ptr) (void)hipFree(*ptr);
  *ptr = nullptr; *have = 0;
  size_t want = need + need / 4;
  hipError_t e = hipMalloc(ptr, want);
  if (e != hipSuccess) return set_error(SG_ERR_HIP, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  *have = want;
  return SG_OK;
}

static int upload_chunked(void *dst, const void *src, size_t bytes) {
  // pageable host memory -> HBM through a double-buffered pinned staging area
  const size_t CH = (size_t)64 << 20;
  if (bytes == 0) return SG_OK;
  if (bytes <= CH) { SHD_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return SG_OK; }
  void *pin[2] = {nullptr, nullptr};
  hipStream_t st;
  SHD_HIP(hipStreamCreate(&st));
  SHD_HIP(hipHostMalloc(&pin[0], CH, hipHostMallocDefault));
  SHD_HIP(hipHostMalloc(&pin[1], CH, hipHostMallocDefault));
  hipEvent_t ev[2];
  SHD_HIP(hipEventCreate(&ev[0])); SHD_HIP(hipEventCreate(&ev[1]));
  int b = 0;
  for (size_t off = 0; off < bytes; off += CH, b ^= 1) {
    size_t len = std::min(CH, bytes - off);
    SHD_HIP(hipEventSynchronize(ev[b]));
    memcpy(pin[b], (const char *)src + off, len);
    SHD_HIP(hipMemcpyAsync((char *)dst + off, pin[b], len, hipMemcpyHostToDevice, st));
    SHD_HIP(hipEventRecord(ev[b], st));
  }
  SHD_HIP(hipStreamSynchronize(st));
  (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
  (void)hipHostFree(pin[0]); (void)hipHostFree(pin[1]);
  (void)hipStreamDestroy(st);
  return SG_OK;
}

extern "C" const char *sg_last_error(void) { return last_error().c_str(); }
// 2: sl_act_norm_* carry (drop_p, drop_seed); gemm / cache / pooling entries.  3: sl_act_norm_* dual (plain + dropped) output
// 4: sg_ppr_push(mode).  5: sl_gat_bwd work buffer holds the datt partial sums.  6: sg_create_from_bin_ex
// 7: sl_spmm_blockdiag_gather_f32, sampler debug entries.  8: sl_sage_fwd / sl_sage_bwd, sl_gemm_pack_b2, sl_gather_rows_drop_f32
extern "C" int sg_abi_version(void) { return 25; }

static int create_common(sg_sampler *s, int device_id, int64_t seed) {
  s->device = device_id;
  s->seed = seed < 0 ? (uint64_t)time(nullptr) : (uint64_t)seed;
  SHD_HIP(hipMalloc((void **)&s->d_counts, (8 * kMaxBatches + 8) * sizeof(uint64_t)));
  SHD_HIP(hipHostMalloc((void **)&s->h_counts, (8 * kMaxBatches + 8) * sizeof(uint64_t), hipHostMallocDefault));
  SHD_HIP(hipEventCreateWithFlags(&s->ev, hipEventDisableTiming));
  return SG_OK;
}

extern "C" int sg_create(const uint32_t *indptr, const uint32_t *indices, uint32_t num_nodes,
                         uint64_t num_edges, int on_device, int device_id, int64_t seed,
                         sg_sampler **out) {
  if (!indptr || (!indices && num_edges) || !out) return set_error(SG_ERR_INVALID, "sg_create: null argument");
  if (num_edges > 0xFFFFFFFFull) return set_error(SG_ERR_INVALID, "sg_create: nnz exceeds uint32 edge ids");
  SHD_HIP(hipSetDevice(device_id));
  sg_sampler *s = new sg_sampler();
  s->N = num_nodes; s->nnz = num_edges;
  int rc;
  if (on_device) {
    s->d_indptr = const_cast<uint32_t *>(indptr);
    s->d_indices = const_cast<uint32_t *>(indices);
    s->owns_graph = false;
  } else {
    if (indptr[0] != 0 || indptr[num_nodes] != num_edges) {     // Graph.h:29-30
      delete s;
      return set_error(SG_ERR_INVALID, "sg_create: indptr[0]=%u indptr[N]=%u but nnz=%llu", indptr[0],
                       indptr[num_nodes], (unsigned long long)num_edges);
    }
    hipError_t e1 = hipMalloc((void **)&s->d_indptr, ((size_t)num_nodes + 1) * 4);
    hipError_t e2 = hipMalloc((void **)&s->d_indices, std::max<size_t>(1, num_edges) * 4);
    if (e1 != hipSuccess || e2 != hipSuccess) { delete s; return set_error(SG_ERR_HIP, "sg_create: hipMalloc of the CSR failed"); }
    s->owns_graph = true;
    if ((rc = upload_chunked(s->d_indptr, indptr, ((size_t)num_nodes + 1) * 4)) != SG_OK) { sg_destroy(s); return rc; }
    if ((rc = upload_chunked(s->d_indices, indices, (size_t)num_edges * 4)) != SG_OK) { sg_destroy(s); return rc; }
  }
  if ((rc = create_common(s, device_id, seed)) != SG_OK) { sg_destroy(s); return rc; }
  *out = s;
  return SG_OK;
}

static int read_bin_u32(const char *path, std::vector<uint32_t> &v) {
  // raw little-endian uint32, element count = file size / 4 (.cpp:81)
  int fd = open(path, O_RDONLY);
  if (fd < 0) return set_error(SG_ERR_IO, "cannot open %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return set_error(SG_ERR_IO, "cannot stat %s", path); }
  size_t n = (size_t)st.st_size / 4;
  v.resize(n);
  size_t got = 0, want = n * 4;
  while (got < want) {
    ssize_t r = read(fd, (char *)v.data() + got, want - got);
    if (r <= 0) { close(fd); return set_error(SG_ERR_IO, "short read on %s", path); }
    got += (size_t)r;
  }
  close(fd);
  return SG_OK;
}

// File -> HBM through the double-buffered pinned staging area, never holding the whole array on the host
// (papers100M: 13.4 GB of CSR).  width = 4: raw uint32 as the reference writes them (ndarray.tofile,
// data_converter.py:462-468, read by .cpp:70-86); width = 8: uint64 / int64 elements (what numpy / scipy hold for
// graphs with more than 2^31 edges when nobody narrowed them), narrowed to uint32 on the way with a range check.
static int stream_file_to_device(uint32_t *dst, int fd, const char *path, size_t count, int width) {
  const size_t CH_ELEMS = (size_t)16 << 20;                       // 64 MiB of uint32 per chunk
  if (count == 0) return SG_OK;
  void *pin[2] = {nullptr, nullptr};
  hipStream_t st;
  hipEvent_t ev[2];
  std::vector<uint64_t> wide;
  if (width == 8) wide.resize(std::min(count, CH_ELEMS));
  SHD_HIP(hipStreamCreate(&st));
  SHD_HIP(hipHostMalloc(&pin[0], std::min(count, CH_ELEMS) * 4, hipHostMallocDefault));
  SHD_HIP(hipHostMalloc(&pin[1], std::min(count, CH_ELEMS) * 4, hipHostMallocDefault));
  SHD_HIP(hipEventCreate(&ev[0])); SHD_HIP(hipEventCreate(&ev[1]));
  int rc = SG_OK, b = 0;
  for (size_t off = 0; off < count && rc == SG_OK; off += CH_ELEMS, b ^= 1) {
    const size_t len = std::min(CH_ELEMS, count - off);
    SHD_HIP(hipEventSynchronize(ev[b]));
    char *to = width == 8 ? (char *)wide.data() : (char *)pin[b];
    size_t got = 0, want = len * (size_t)width;
    while (got < want) {
      ssize_t r = pread(fd, to + got, want - got, (off_t)(off * (size_t)width + got));
      if (r <= 0) { rc = set_error(SG_ERR_IO, "short read on %s", path); break; }
      got += (size_t)r;
    }
    if (rc != SG_OK) break;
    if (width == 8) {
      uint32_t *narrow = (uint32_t *)pin[b];
      uint64_t over = 0;
      for (size_t i = 0; i < len; i++) { over |= wide[i] >> 32; narrow[i] = (uint32_t)wide[i]; }
      if (over) { rc = set_error(SG_ERR_INVALID, "%s holds values >= 2^32: not representable in the uint32 node / edge ids "
                                 "of the sampler (Graph.h:16)", path); break; }
    }
    SHD_HIP(hipMemcpyAsync(dst + off, pin[b], len * 4, hipMemcpyHostToDevice, st));
    SHD_HIP(hipEventRecord(ev[b], st));
  }
  (void)hipStreamSynchronize(st);
  (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
  (void)hipHostFree(pin[0]); (void)hipHostFree(pin[1]);
  (void)hipStreamDestroy(st);
  return rc;
}

static int file_elems(const char *path, int width, int *fd_out, size_t *count) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) return set_error(SG_ERR_IO, "cannot open %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return set_error(SG_ERR_IO, "cannot stat %s", path); }
  *count = (size_t)st.st_size / (size_t)width;                    // element count = file size / element size (.cpp:81)
  *fd_out = fd;
  return SG_OK;
}

static int read_elem(int fd, const char *path, size_t index, int width, uint64_t *v) {
  uint64_t x = 0;
  if (pread(fd, &x, (size_t)width, (off_t)(index * (size_t)width)) != (ssize_t)width) return set_error(SG_ERR_IO, "short read on %s", path);
  *v = width == 8 ? x : (uint64_t)(uint32_t)x;
  return SG_OK;
}

extern "C" int sg_create_from_bin_ex(const char *path_indptr, const char *path_indices, int indptr_bytes, int indices_bytes,
                                     int device_id, int64_t seed, sg_sampler **out) {
  if (!path_indptr || !path_indices || !out) return set_error(SG_ERR_INVALID, "sg_create_from_bin: null argument");
  if ((indptr_bytes != 4 && indptr_bytes != 8) || (indices_bytes != 4 && indices_bytes != 8))
    return set_error(SG_ERR_INVALID, "sg_create_from_bin: element widths must be 4 or 8 bytes");
  int fd_p = -1, fd_x = -1, rc;
  size_t np = 0, nx = 0;
  if ((rc = file_elems(path_indptr, indptr_bytes, &fd_p, &np)) != SG_OK) return rc;
  if ((rc = file_elems(path_indices, indices_bytes, &fd_x, &nx)) != SG_OK) { close(fd_p); return rc; }
  auto fail = [&](int code) { close(fd_p); close(fd_x); return code; };
  if (np == 0) return fail(set_error(SG_ERR_IO, "%s is empty", path_indptr));
  if (np - 1 > 0xFFFFFFFFull) return fail(set_error(SG_ERR_INVALID, "%s: more than 2^32 nodes", path_indptr));
  uint64_t first = 0, last = 0;
  if ((rc = read_elem(fd_p, path_indptr, 0, indptr_bytes, &first)) != SG_OK) return fail(rc);
  if ((rc = read_elem(fd_p, path_indptr, np - 1, indptr_bytes, &last)) != SG_OK) return fail(rc);
  if (nx > 0xFFFFFFFFull || last > 0xFFFFFFFFull)
    return fail(set_error(SG_ERR_INVALID, "%s: %llu edges do not fit the sampler's uint32 edge ids (Graph.h:16, "
                          "ParallelSampler.h: vector<NodeType> origEdgeID)", path_indices, (unsigned long long)std::max<uint64_t>(nx, last)));
  if (first != 0 || last != nx)                                    // Graph.h:29-30
    return fail(set_error(SG_ERR_INVALID, "sg_create_from_bin: indptr[0]=%llu indptr[N]=%llu but %s holds %llu ids",
                          (unsigned long long)first, (unsigned long long)last, path_indices, (unsigned long long)nx));
  SHD_HIP(hipSetDevice(device_id));
  sg_sampler *s = new sg_sampler();
  s->N = (uint32_t)(np - 1); s->nnz = nx; s->owns_graph = true;
  hipError_t e1 = hipMalloc((void **)&s->d_indptr, np * 4);
  hipError_t e2 = hipMalloc((void **)&s->d_indices, std::max<size_t>(1, nx) * 4);
  if (e1 != hipSuccess || e2 != hipSuccess) { sg_destroy(s); return fail(set_error(SG_ERR_HIP, "sg_create_from_bin: hipMalloc of the CSR failed")); }
  if ((rc = stream_file_to_device(s->d_indptr, fd_p, path_indptr, np, indptr_bytes)) != SG_OK) { sg_destroy(s); return fail(rc); }
  if ((rc = stream_file_to_device(s->d_indices, fd_x, path_indices, nx, indices_bytes)) != SG_OK) { sg_destroy(s); return fail(rc); }
  close(fd_p); close(fd_x);
  if ((rc = create_common(s, device_id, seed)) != SG_OK) { sg_destroy(s); return rc; }
  *out = s;
  return SG_OK;
}

extern "C" int sg_create_from_bin(const char *path_indptr, const char *path_indices, int device_id,
                                  int64_t seed, sg_sampler **out) {
  return sg_create_from_bin_ex(path_indptr, path_indices, 4, 4, device_id, seed, out);
}

static void free_ppr(sg_sampler *s) {
  if (s->d_ppr_row) (void)hipFree(s->d_ppr_row);
  if (s->d_ppr_len) (void)hipFree(s->d_ppr_len);
  if (s->d_ppr_neigh) (void)hipFree(s->d_ppr_neigh);
  if (s->d_ppr_score) (void)hipFree(s->d_ppr_score);
  s->d_ppr_row = nullptr; s->d_ppr_len = nullptr; s->d_ppr_neigh = nullptr; s->d_ppr_score = nullptr;
  s->ppr_rows = 0; s->ppr_stride = 0;
}

extern "C" void sg_destroy(sg_sampler *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->pending && s->ev) (void)hipEventSynchronize(s->ev);
  if (s->owns_graph) { if (s->d_indptr) (void)hipFree(s->d_indptr); if (s->d_indices) (void)hipFree(s->d_indices); }
  if (s->d_targets) (void)hipFree(s->d_targets);
  if (s->d_self_slot) (void)hipFree(s->d_self_slot);
  free_ppr(s);
  if (s->d_scratch) (void)hipFree(s->d_scratch);
  if (s->d_big) (void)hipFree(s->d_big);
  if (s->d_counts) (void)hipFree(s->d_counts);
  if (s->h_counts) (void)hipHostFree(s->h_counts);
  if (s->ev) (void)hipEventDestroy(s->ev);
  for (int i = 0; i < 3; i++) if (s->ev_t[i]) (void)hipEventDestroy(s->ev_t[i]);
  delete s;
}

extern "C" uint32_t sg_num_nodes(const sg_sampler *s) { return s ? s->N : 0; }
extern "C" uint64_t sg_num_edges(const sg_sampler *s) { return s ? (s->graph_dropped ? 0 : s->nnz) : 0; }
extern "C" uint64_t sg_num_nodes_target(const sg_sampler *s) { return s ? s->num_targets : 0; }
extern "C" uint64_t sg_get_idx_root(const sg_sampler *s) { return s ? s->idx_root : 0; }
extern "C" const uint32_t *sg_device_indptr(const sg_sampler *s) { return s ? s->d_indptr : nullptr; }
extern "C" const uint32_t *sg_device_indices(const sg_sampler *s) { return s ? s->d_indices : nullptr; }

extern "C" int sg_shuffle_targets(sg_sampler *s, const uint32_t *h_targets, uint64_t count) {
  if (!s || (!h_targets && count)) return set_error(SG_ERR_INVALID, "sg_shuffle_targets: null argument");
  SHD_HIP(hipSetDevice(s->device));
  for (uint64_t i = 0; i < count; i++)
    if (h_targets[i] >= s->N) return set_error(SG_ERR_INVALID, "sg_shuffle_targets: target %u >= num_nodes %u", h_targets[i], s->N);
  if (s->pending) SHD_HIP(hipEventSynchronize(s->ev));
  if (count > s->cap_targets) {
    if (s->d_targets) (void)hipFree(s->d_targets);
    s->d_targets = nullptr; s->cap_targets = 0;
    SHD_HIP(hipMalloc((void **)&s->d_targets, std::max<uint64_t>(1, count) * 4));
    s->cap_targets = count;
  }
  if (count) SHD_HIP(hipMemcpy(s->d_targets, h_targets, count * 4, hipMemcpyHostToDevice));
  s->num_targets = count;
  return SG_OK;
}

extern "C" int sg_next_roots(sg_sampler *s, uint32_t num_roots, uint32_t max_subgraphs,
                             uint64_t *root_start, uint32_t *num_subgraphs, uint64_t *serial_base) {
  if (!s || !root_start || !num_subgraphs || !serial_base || num_roots == 0)
    return set_error(SG_ERR_INVALID, "sg_next_roots: bad argument");
  if (s->num_targets == 0) return set_error(SG_ERR_STATE, "sg_next_roots: no targets (call sg_shuffle_targets)");
  // _get_roots_p, .cpp:459-468
  // (a shorter target list installed mid-epoch leaves the cursor past its end: wrap, as at an epoch end --
  //  the reference would index past `targets` there)
  const uint64_t start = s->idx_root < s->num_targets ? s->idx_root : 0;
  const uint64_t end = std::min<uint64_t>(s->num_targets, start + (uint64_t)num_roots * max_subgraphs);
  s->idx_root = (end == s->num_targets) ? 0 : end;
  const uint64_t groups = (end - start + num_roots - 1) / num_roots;
  if ((end - start) % num_roots != 0)
    return set_error(SG_ERR_STATE, "sg_next_roots: %llu remaining targets not divisible by num_roots=%u",
                     (unsigned long long)(end - start), num_roots);
  *root_start = start;
  *num_subgraphs = (uint32_t)groups;
  *serial_base = s->serial_next;
  s->serial_next += groups;
  return SG_OK;
}

extern "C" int sg_set_caps(sg_sampler *s, uint32_t cap_subg_nodes, uint32_t cap_subg_edges) {
  if (!s) return set_error(SG_ERR_INVALID, "sg_set_caps: null sampler");
  s->user_cap_nodes = std::max(s->user_cap_nodes, cap_subg_nodes);
  s->user_cap_edges = std::max(s->user_cap_edges, cap_subg_edges);
  return SG_OK;
}

static void derive_caps(const sg_sampler *s, const sg_config *cfg, uint32_t *capn, uint32_t *cape,
                        uint32_t *capf) {
  const uint64_t N = std::max<uint32_t>(1, s->N);
  uint64_t n = 1, f = 1;
  const uint64_t R = (uint64_t)std::max(1, cfg->num_roots);
  if (cfg->method == SG_METHOD_KHOP) {
    if (cfg->budget >= 0) {
      uint64_t lvl = R; n = R; f = R;
      for (int d = 0; d < cfg->depth; d++) {
        lvl = std::min<uint64_t>(N, lvl * (uint64_t)cfg->budget);
        if (d + 1 < cfg->depth) f = std::max(f, lvl);
        n = std::min<uint64_t>(N, n + lvl);
      }
    } else {
      n = std::min<uint64_t>(N, 4096);   // grown on demand (SG_ERR_CAPACITY)
      f = n;
    }
  } else if (cfg->method == SG_METHOD_PPR) {
    n = std::min<uint64_t>(N, R * ((uint64_t)std::max(0, cfg->k) + 1)); f = 1;
  } else {
    n = R; f = R;
  }
  // grown capacities (sg_set_caps after SG_ERR_CAPACITY) only concern k-hop: the other methods are bounded
  if (cfg->method == SG_METHOD_KHOP) n = std::max<uint64_t>(n, s->user_cap_nodes);
  n = std::min<uint64_t>(n, N);
  n = std::max<uint64_t>(n, R);
  if (cfg->method == SG_METHOD_KHOP && cfg->budget < 0) f = n;
  f = std::min<uint64_t>(std::max<uint64_t>(f, R), n);
  uint64_t e = std::min<uint64_t>(n * n + n, std::max<uint64_t>(4096, 16 * n));
  if (cfg->method == SG_METHOD_KHOP) e = std::max<uint64_t>(e, s->user_cap_edges);
  e = std::min<uint64_t>(e, 0x7FFFFFFFull);
  *capn = (uint32_t)n; *cape = (uint32_t)e; *capf = (uint32_t)f;
}

extern "C" int sg_get_caps(const sg_sampler *s, const sg_config *cfg, uint32_t *cap_subg_nodes,
                           uint32_t *cap_subg_edges) {
  if (!s || !cfg || !cap_subg_nodes || !cap_subg_edges) return set_error(SG_ERR_INVALID, "sg_get_caps: null argument");
  uint32_t f;
  derive_caps(s, cfg, cap_subg_nodes, cap_subg_edges, &f);
  return SG_OK;
}

extern "C" int sg_set_ppr(sg_sampler *s, const uint32_t *h_targets, uint32_t num_rows,
                          const uint32_t *h_len, const uint32_t *h_neigh, const float *h_score,
                          uint32_t stride) {
  if (!s || (num_rows && (!h_targets || !h_len || !h_neigh || !h_score)) || stride == 0)
    return set_error(SG_ERR_INVALID, "sg_set_ppr: bad argument");
  SHD_HIP(hipSetDevice(s->device));
  if (s->pending) SHD_HIP(hipEventSynchronize(s->ev));
  free_ppr(s);
  std::vector<int32_t> row((size_t)s->N, -1);
  for (uint32_t r = 0; r < num_rows; r++) {
    if (h_targets[r] >= s->N) return set_error(SG_ERR_INVALID, "sg_set_ppr: target %u out of range", h_targets[r]);
    if (h_len[r] > stride) return set_error(SG_ERR_INVALID, "sg_set_ppr: row %u longer than stride", r);
    row[h_targets[r]] = (int32_t)r;
  }
  const size_t cells = std::max<size_t>(1, (size_t)num_rows * stride);
  SHD_HIP(hipMalloc((void **)&s->d_ppr_row, std::max<size_t>(1, s->N) * 4));
  SHD_HIP(hipMalloc((void **)&s->d_ppr_len, std::max<size_t>(1, num_rows) * 4));
  SHD_HIP(hipMalloc((void **)&s->d_ppr_neigh, cells * 4));
  SHD_HIP(hipMalloc((void **)&s->d_ppr_score, cells * 4));
  int rc;
  if ((rc = upload_chunked(s->d_ppr_row, row.data(), (size_t)s->N * 4)) != SG_OK) return rc;
  if (num_rows) {
    if ((rc = upload_chunked(s->d_ppr_len, h_len, (size_t)num_rows * 4)) != SG_OK) return rc;
    if ((rc = upload_chunked(s->d_ppr_neigh, h_neigh, (size_t)num_rows * stride * 4)) != SG_OK) return rc;
    if ((rc = upload_chunked(s->d_ppr_score, h_score, (size_t)num_rows * stride * 4)) != SG_OK) return rc;
  }
  s->ppr_rows = num_rows; s->ppr_stride = stride;
  s->h_ppr_targets.assign(h_targets, h_targets + num_rows);
  s->h_ppr_len.assign(h_len, h_len + num_rows);
  s->h_ppr_neigh.assign(h_neigh, h_neigh + (size_t)num_rows * stride);
  s->h_ppr_score.assign(h_score, h_score + (size_t)num_rows * stride);
  return SG_OK;
}

// PPR cache files: header {float alpha'(=1-alpha), float epsilon, int32 k, uint32 num_records},
// then per node {uint32 len, len * (uint32 | float)}  (.cpp:105-137).
extern "C" int sg_save_ppr_bin(const sg_sampler *s, const char *path_neighs, const char *path_scores,
                               int k, float alpha, float epsilon) {
  if (!s || !path_neighs || !path_scores) return set_error(SG_ERR_INVALID, "sg_save_ppr_bin: null argument");
  if (!s->ppr_rows) return set_error(SG_ERR_STATE, "sg_save_ppr_bin: no PPR table set");
  const float a1 = 1 - alpha;
  std::vector<int32_t> row((size_t)s->N, -1);
  for (uint32_t r = 0; r < s->ppr_rows; r++) row[s->h_ppr_targets[r]] = (int32_t)r;
  for (int which = 0; which < 2; which++) {
    FILE *f = fopen(which ? path_scores : path_neighs, "wb");
    if (!f) return set_error(SG_ERR_IO, "cannot write %s", which ? path_scores : path_neighs);
    uint32_t cnt = s->N;
    fwrite(&a1, 4, 1, f); fwrite(&epsilon, 4, 1, f); fwrite(&k, 4, 1, f); fwrite(&cnt, 4, 1, f);
    for (uint32_t v = 0; v < s->N; v++) {
      uint32_t len = row[v] < 0 ? 0u : s->h_ppr_len[row[v]];
      fwrite(&len, 4, 1, f);
      if (len) {
        const size_t off = (size_t)row[v] * s->ppr_stride;
        if (which) fwrite(s->h_ppr_score.data() + off, 4, len, f);
        else fwrite(s->h_ppr_neigh.data() + off, 4, len, f);
      }
    }
    fclose(f);
  }
  return SG_OK;
}

extern "C" int sg_load_ppr_bin(sg_sampler *s, const char *path_neighs, const char *path_scores, int k,
                               float alpha, float epsilon) {
  if (!s || !path_neighs || !path_scores || k <= 0) return set_error(SG_ERR_INVALID, "sg_load_ppr_bin: bad argument");
  const float a1 = 1 - alpha;
  std::vector<uint32_t> raw[2];
  int rc;
  if ((rc = read_bin_u32(path_neighs, raw[0])) != SG_OK) return rc;
  if ((rc = read_bin_u32(path_scores, raw[1])) != SG_OK) return rc;
  for (int w = 0; w < 2; w++) {
    if (raw[w].size() < 4) return set_error(SG_ERR_IO, "PPR file too short");
    float a_, e_; int32_t k_;
    memcpy(&a_, &raw[w][0], 4); memcpy(&e_, &raw[w][1], 4); memcpy(&k_, &raw[w][2], 4);
    // acceptance rule of read_PPR_from_binary_file, .cpp:166
    if (a_ != a1 || e_ > 1.1 * epsilon || e_ < 0.9 * epsilon || k_ < k)
      return set_error(SG_ERR_IO, "PPR file header mismatch (alpha'=%g eps=%g k=%d)", a_, e_, k_);
    if (raw[w][3] != s->N) return set_error(SG_ERR_IO, "PPR file has %u records, graph has %u nodes", raw[w][3], s->N);
  }
  std::vector<uint32_t> targets, len, neigh;
  std::vector<float> score;
  size_t o0 = 4, o1 = 4;
  for (uint32_t v = 0; v < s->N; v++) {
    if (o0 >= raw[0].size() || o1 >= raw[1].size()) return set_error(SG_ERR_IO, "PPR file truncated");
    const uint32_t l0 = raw[0][o0++], l1 = raw[1][o1++];
    if (l0 != l1 || o0 + l0 > raw[0].size() || o1 + l1 > raw[1].size()) return set_error(SG_ERR_IO, "PPR files inconsistent");
    if (l0) {
      const uint32_t clip = std::min<uint32_t>(l0, (uint32_t)k);      // .cpp:176-183
      targets.push_back(v); len.push_back(clip);
      const size_t base = neigh.size();
      neigh.resize(base + k, 0xFFFFFFFFu); score.resize(base + k, 0.0f);
      for (uint32_t j = 0; j < clip; j++) { neigh[base + j] = raw[0][o0 + j]; memcpy(&score[base + j], &raw[1][o1 + j], 4); }
    }
    o0 += l0; o1 += l1;
  }
  return sg_set_ppr(s, targets.data(), (uint32_t)targets.size(), len.data(), neigh.data(), score.data(), (uint32_t)k);
}

extern "C" int sg_drop_full_graph_info(sg_sampler *s) {
  if (!s) return set_error(SG_ERR_INVALID, "sg_drop_full_graph_info: null sampler");
  SHD_HIP(hipSetDevice(s->device));
  if (s->pending) SHD_HIP(hipEventSynchronize(s->ev));
  if (s->owns_graph && s->d_indices) { (void)hipFree(s->d_indices); s->d_indices = nullptr; }
  free_ppr(s);
  s->graph_dropped = true;
  return SG_OK;
}

static int sample_impl(sg_sampler *s, const sg_config *cfg, uint64_t root_start, uint32_t nbatch, const uint32_t *bsize,
                       uint64_t serial_base, const uint32_t *d_roots_override, const sg_batch_out *outs, void *stream_) {
  if (!s || !cfg || !outs || !bsize) return set_error(SG_ERR_INVALID, "sg_sample: null argument");
  if (nbatch < 1 || nbatch > kMaxBatches) return set_error(SG_ERR_INVALID, "sg_sample_multi: %u batches per call (1..%u)", nbatch, kMaxBatches);
  hipStream_t stream = (hipStream_t)stream_;
  uint64_t Psum = 0;
  for (uint32_t b = 0; b < nbatch; b++) {
    if (nbatch > 1 && bsize[b] == 0) return set_error(SG_ERR_INVALID, "sg_sample_multi: batch %u is empty", b);
    Psum += bsize[b];
  }
  if (Psum >= ((uint64_t)1 << 31)) return set_error(SG_ERR_INVALID, "sg_sample: too many subgraphs in one call");
  const uint32_t P = (uint32_t)Psum;
  const int R = cfg->num_roots;
  if (s->graph_dropped) return set_error(SG_ERR_STATE, "sg_sample: full graph was dropped");
  if (R < 1 || R > (int)kMaxRoots) return set_error(SG_ERR_INVALID, "sg_sample: num_roots=%d unsupported (1..%u)", R, kMaxRoots);
  if (cfg->method < 0 || cfg->method > SG_METHOD_NODEIID) return set_error(SG_ERR_INVALID, "sg_sample: unknown method %d", cfg->method);
  if (cfg->method == SG_METHOD_KHOP && (cfg->depth < 0 || cfg->depth > 30)) return set_error(SG_ERR_INVALID, "sg_sample: depth=%d unsupported (0..30)", cfg->depth);
  if (cfg->method == SG_METHOD_KHOP && cfg->budget > 262144) return set_error(SG_ERR_INVALID, "sg_sample: budget too large");
  if (cfg->method == SG_METHOD_PPR && !s->d_ppr_row) return set_error(SG_ERR_STATE, "sg_sample: PPR table not set (sg_set_ppr / sg_load_ppr_bin)");
  if (cfg->method == SG_METHOD_PPR && cfg->k < 0) return set_error(SG_ERR_INVALID, "sg_sample: k<0");
  if ((cfg->aug_flags & SG_AUG_DRNLS) && R < 2) return set_error(SG_ERR_INVALID, "sg_sample: drnl needs two roots");
  if (!d_roots_override) {
    if (root_start + (uint64_t)P * R > s->num_targets)
      return set_error(SG_ERR_INVALID, "sg_sample: roots [%llu, +%u*%d) outside the %llu targets",
                       (unsigned long long)root_start, P, R, (unsigned long long)s->num_targets);
  }
  for (uint32_t b = 0; b < nbatch; b++) {
    const sg_batch_out *out = outs + b;
    if (!out->d_node || !out->d_indptr || !out->d_indices || !out->d_edge_id || !out->d_target ||
        !out->d_subg_nodes || !out->d_subg_edges)
      return set_error(SG_ERR_INVALID, "sg_sample: missing output buffer (batch %u)", b);
  }
  const sg_batch_out *out = outs;
  SHD_HIP(hipSetDevice(s->device));
  if (s->pending) { SHD_HIP(hipEventSynchronize(s->ev)); s->pending = false; }

  uint32_t capn, cape, capf;
  derive_caps(s, cfg, &capn, &cape, &capf);
  // scratch carve
  const size_t Pz = std::max<uint32_t>(1, P);
  size_t o = 0;
  auto carve = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
  const size_t o_nodes = carve(Pz * capn * 4), o_ppr = carve(Pz * capn * 4), o_rowptr = carve(Pz * ((size_t)capn + 1) * 4);
  const size_t o_tmp = carve(Pz * capn * 4);
  const size_t o_row = carve(Pz * (size_t)cape * 4);
  const size_t o_col = carve(Pz * (size_t)cape * 4), o_eid = carve(Pz * (size_t)cape * 4);
  const size_t o_tgt = carve(Pz * kMaxRoots * 4), o_cnt = carve(Pz * R_WORDS * 4);
  const size_t o_info = carve(Pz * capn * sizeof(RowInfo)), o_rowq = carve(Pz * ((size_t)capn + 1) * 4);
  const size_t o_lcol = carve((cfg->aug_flags & (SG_AUG_HOPS | SG_AUG_DRNLS)) ? Pz * (size_t)cape * 4 : 16);
  // (flat scan with self-edge insertion: the rows' self-edge slots, written beside the row records -- see SampleParams::s_selfpos)
  const bool self_edges = cfg->method != SG_METHOD_NODEIID && cfg->add_self_edge;
  const size_t o_selfpos = carve(self_edges ? Pz * capn * 4 : 16);
  const uint32_t kScanGridMax = 8u * 256u;
  const uint32_t rec_blocks = (2u * kScanGridMax + (uint32_t)Pz / 4u) * s->rec_scale;
  const size_t o_cstart = carve((Pz + 1) * 4), o_plan = carve(PL_WORDS * 4);
  const size_t o_recs = carve((size_t)rec_blocks * kRecPerBlock * sizeof(RoundRec)), o_blkinfo = carve((size_t)rec_blocks * sizeof(uint2));
  int rc;
  if ((rc = ensure(&s->d_scratch, &s->scratch_bytes, o)) != SG_OK) return rc;
  char *sc = (char *)s->d_scratch;

  s->timed = false;
  SHD_HIP(hipMemsetAsync(s->d_counts, 0, (8 * kMaxBatches + 8) * sizeof(uint64_t), stream));
  if (P == 0) {
    SHD_HIP(hipMemsetAsync(out->d_indptr, 0, 4, stream));
    SHD_HIP(hipMemsetAsync(out->d_subg_nodes, 0, 4, stream));
    SHD_HIP(hipMemsetAsync(out->d_subg_edges, 0, 4, stream));
  } else {
    if (s->profiling) SHD_HIP(hipEventRecord(s->ev_t[0], stream));
    SampleParams p;
    memset(&p, 0, sizeof(p));
    p.indptr = s->d_indptr; p.indices = s->d_indices; p.N = s->N; p.nnz = s->nnz;
    p.roots = d_roots_override ? d_roots_override : s->d_targets + root_start;
    p.P = P; p.R = R;
    p.method = cfg->method; p.depth = cfg->depth; p.budget = cfg->budget; p.k = cfg->k;
    p.threshold = cfg->threshold;
    p.include_self = (cfg->method == SG_METHOD_NODEIID) ? 0 : cfg->add_self_edge;   // .cpp:506
    p.include_target_conn = (cfg->method == SG_METHOD_NODEIID) ? 0 : cfg->include_target_conn;
    p.compat = cfg->compat_overread;
    p.seed = s->seed; p.serial_base = serial_base;
    p.ppr_row = s->d_ppr_row; p.ppr_len = s->d_ppr_len; p.ppr_neigh = s->d_ppr_neigh;
    p.ppr_score = s->d_ppr_score; p.ppr_stride = s->ppr_stride;
    p.cap_nodes_scr = capn; p.cap_edges_scr = cape;
    p.s_nodes = (uint32_t *)(sc + o_nodes); p.s_ppr = (float *)(sc + o_ppr);
    p.s_row = (uint32_t *)(sc + o_row); p.s_col = (uint32_t *)(sc + o_col);
    p.s_eid = (uint32_t *)(sc + o_eid); p.s_tgt = (uint32_t *)(sc + o_tgt);
    p.s_cnt = (uint32_t *)(sc + o_cnt);
    p.s_rowinfo = (RowInfo *)(sc + o_info); p.s_rowq = (uint32_t *)(sc + o_rowq);
    p.cstart = (uint32_t *)(sc + o_cstart); p.plan = (uint32_t *)(sc + o_plan);
    p.recs = (RoundRec *)(sc + o_recs); p.blkinfo = (uint2 *)(sc + o_blkinfo); p.rec_blocks = rec_blocks;
    s->last_cnt = p.s_cnt; s->last_plan = p.plan;
    // The flat scan takes every call without compat over-read whose subgraphs have one root (or include_target_conn); with
    // self-edge insertion it reads each row's slot from s_selfpos, which the SELECTION kernels fill from the per-node table
    // (built once per handle) while they write the row records.
    const char *impl_env0 = getenv("SHADOW_SG_SCAN_IMPL");
    const bool flat_self = p.include_self && !p.compat && (p.include_target_conn || R == 1) && !(impl_env0 && !strcmp(impl_env0, "window"));
    if (flat_self) {
      if (!s->d_self_slot) {
        // once per handle: lower_bound == upper_bound of every node in its own row (ParallelSampler.cpp:386-400), 4 N bytes
        hipError_t e = hipMalloc((void **)&s->d_self_slot, (size_t)s->N * 4);
        if (e != hipSuccess) { s->d_self_slot = nullptr; return set_error(SG_ERR_HIP, "sg_sample: hipMalloc of %zu bytes for the self-edge slots failed", (size_t)s->N * 4); }
        hipLaunchKernelGGL(sg_self_slot_kernel, dim3((s->N + 255) / 256), dim3(256), 0, stream, s->d_indptr, s->d_indices, s->N, s->d_self_slot);
        SHD_HIP(hipGetLastError());
      }
      p.self_slot = s->d_self_slot;
      p.s_selfpos = (uint32_t *)(sc + o_selfpos);
    }
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, s->device);
    auto env_u32 = [](const char *name, uint32_t dflt) { const char *e = getenv(name); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : dflt; };
    SHD_HIP(hipMemsetAsync(p.plan + 20, 0, 4 * sizeof(uint32_t), stream));
    // ---- 1. selection (one workgroup per subgraph, persistent over a ticket; tables in LDS)
    const uint32_t capn_lds = std::min(capn, kLdsCapNodes);
    const uint32_t capf_lds = std::min(capf, capn_lds);
    {
      const uint32_t T = 256;
      // bucketised table: ~4 key slots per possible node (power-of-two bucket count)
      uint32_t H = std::max<uint32_t>(64, next_pow2((uint64_t)capn_lds * 4));
      while ((size_t)H * 8 > 16 * 1024 && (H >> 1) >= capn_lds * 2) H >>= 1;
      p.capn = capn_lds; p.capf = capf_lds; p.H = H;
      p.hshift = 32; for (uint32_t h = H / 4; h > 1; h >>= 1) p.hshift--;
      p.g_ticket = (uint32_t *)(s->d_counts + 8 * kMaxBatches);
      p.lds_skip = 0;
      if (cfg->method == SG_METHOD_KHOP && cfg->budget > 1 && cfg->depth >= 1 && capn > capn_lds) {
        // level sizes of a budgeted expansion: 1 + b + b^2 + ... (an ordinary subgraph reaches about half of it)
        uint64_t est = 1, lvl = 1;
        for (int l = 0; l < cfg->depth && est < ((uint64_t)1 << 40); l++) { lvl *= (uint64_t)cfg->budget; est += lvl; }
        p.lds_skip = est >= (uint64_t)4 * capn_lds ? 1u : 0u;
      }
      const LdsLayout L = lds_layout(H, capn_lds, capf_lds, cfg->method == SG_METHOD_PPR);
      if (L.total > 160 * 1024 - 256) return set_error(SG_ERR_INVALID, "sg_sample: LDS layout %zu B too large", L.total);
      uint32_t per_cu = (uint32_t)std::min<size_t>((size_t)(160 * 1024) / (L.total + 64), 32 / (T / 64));
      per_cu = std::max<uint32_t>(1, std::min<uint32_t>(per_cu, 8));
      const uint32_t grid = std::min<uint32_t>(P, (uint32_t)ncu * per_cu);
      if (L.total > 64 * 1024)
        SHD_HIP(ensure_dynamic_lds((const void *)sg_select_lds_kernel, L.total));
      hipLaunchKernelGGL(sg_select_lds_kernel, dim3(grid), dim3(T), L.total, stream, p);
      SHD_HIP(hipGetLastError());
    }
    // ---- 1b. node sets beyond the LDS tables: the same selection over global-memory tables
    if (capn > capn_lds) {
      SampleParams q = p;
      q.capn = capn; q.capf = capf;
      const uint32_t Tb = 1024;
      uint32_t Hb = std::max<uint32_t>(64, next_pow2((uint64_t)capn * 4));
      q.H = Hb; q.hshift = 32; for (uint32_t h = Hb / 4; h > 1; h >>= 1) q.hshift--;
      uint64_t stride = ((uint64_t)Hb + kStash) * 3 + ((uint64_t)capn + 8) + ((uint64_t)capf + 4) * 2 + 64;
      stride = (stride + 63) & ~(uint64_t)63;
      uint32_t nslots = std::min<uint32_t>(P, (uint32_t)ncu);
      // keep the table arena below 4 GiB
      while (nslots > 1 && (uint64_t)nslots * stride * 4 > ((uint64_t)4 << 30)) nslots >>= 1;
      if ((rc = ensure(&s->d_big, &s->big_bytes, (size_t)nslots * stride * 4)) != SG_OK) return rc;
      q.g_tables = (uint32_t *)s->d_big; q.g_stride = stride;
      q.g_ticket = (uint32_t *)(s->d_counts + 8 * kMaxBatches) + 1;
      // room to sort a subgraph's ids in LDS (ids + the counting sort's scratch): up to ~18 k nodes
      const size_t sort_bytes = ((size_t)2 * capn + 768) * 4;
      q.capm = sort_bytes <= (size_t)150 * 1024 ? capn : 0u;           // (capm is a scan parameter: free in this launch)
      const size_t big_lds = q.capm ? sort_bytes : 0;
      if (big_lds > 48 * 1024) SHD_HIP(ensure_dynamic_lds((const void *)sg_select_big_kernel, big_lds));
      hipLaunchKernelGGL(sg_select_big_kernel, dim3(nslots), dim3(Tb), big_lds, stream, q);
      SHD_HIP(hipGetLastError());
    }
    // ---- 2. plan (chunk prefix, chunks per scan workgroup) + 3. scan: one equal span of the chunk sequence per workgroup
    {
      const bool big = capn > kLdsCapNodes;
      if (capn >= (1u << kRankShift)) return set_error(SG_ERR_INVALID, "sg_sample: subgraphs of %u nodes exceed the scan's row field", capn);
      p.bit_words = env_u32("SHADOW_SG_BITWORDS", big ? kBitWordsBig : kBitWords);
      if (p.bit_words & (p.bit_words - 1)) p.bit_words = big ? kBitWordsBig : kBitWords;
      p.nodes_lds = std::min(capn, big ? 1024u : kLdsCapNodes);
      // (node sets beyond the LDS tables: one workgroup per CU -- 16 wavefronts keep twice the loads in flight: 1.59 -> 1.33 ms
      //  at 256 depth-3 roots)
      const bool plain = !p.include_self && !p.compat && (p.include_target_conn || R == 1);
      // SHADOW_SG_SCAN_IMPL=window: the general (row-window) kernel for plain calls too (A/B measurements, tests)
      // (round 5: the flat kernel also takes calls WITH self-edge insertion -- the streaming loop finds the slot where the ids
      //  pass by; compat over-read and the root<->root exclusion stay with the row-window kernel)
      const char *impl_env = getenv("SHADOW_SG_SCAN_IMPL");
      const bool flat = !p.compat && (p.include_target_conn || R == 1) && !(impl_env && !strcmp(impl_env, "window"));
      const uint32_t Tdef = (big && flat) ? 1024u : 512u;
      uint32_t T = env_u32("SHADOW_SG_SCAN_THREADS", Tdef);
      if (T != 256 && T != 512 && T != 1024) T = Tdef;
      // candidate list: about one id in a hundred of a round.  The flat kernel trades 512 entries for a longer run list --
      // whole subgraphs then fit one round (scripts/sweep_capm.sh: 23 % fewer rounds, 0.205 -> 0.200 ms at 1 024 roots,
      // 1.105 -> 1.047 ms at 8 192); a round that overflows the list is redone on half the quads.
      p.capm = std::min<uint32_t>(4096, std::max<uint32_t>(1024, env_u32("SHADOW_SG_CAPM", big ? 1024 : (flat ? 1536 : 2048))));
      if (flat && big && !getenv("SHADOW_SG_BITWORDS") && !getenv("SHADOW_SG_CAPM")) {
        // Node sets beyond the 2 048-node tables on the flat kernel (depth-3 k-hop: ~4 600 nodes, 590 chunks per subgraph): a
        // round costs ~5 us of dependent round trips whatever it streams, so the LDS goes to LONG rounds -- 1 024 runs and as
        // many candidates as fit -- and to the whole sorted node list (candidates then resolve without leaving the CU)
        // instead of a 1 Mi-bit filter: 512 Ki bits hold a 4 600-node set at 0.9 % false positives (one more candidate
        // per true one).  Measured at 256 depth-3 roots: 22.7 -> 8.1 rounds per segment, scan 1.18 -> 0.8 ms.
        p.bit_words = capn <= 16384 ? 16384u : kBitWordsBig;
        p.nodes_lds = (size_t)capn * 4 <= 36 * 1024 ? capn : 1024u;
        const size_t fixed = scan_layout(p.bit_words, 0, p.nodes_lds, T / 64, 0u, 1024, p.include_self != 0).total + kHubCap * 32;
        const size_t room = (size_t)160 * 1024 - 256 - 64;
        if (fixed + 1024 * 16 <= room) p.capm = std::min<uint32_t>(4096, (uint32_t)((room - fixed) / 16) & ~63u);
      }
      p.run_cap = 0;
      p.seg_pad = std::min<uint32_t>(256, env_u32("SHADOW_SG_SEG_PAD", 25) - 1);   // (env value = pad + 1: 1 means none; default 24, scripts/sweep_seg_pad.sh)
      if (flat) {
        // the run list takes what two workgroups per CU leave (one per CU when the filter alone is larger)
        const size_t per_run = p.include_self ? 20 : 16;           // (run record + the row's id for self-edge insertion)
        const size_t base = scan_layout(p.bit_words, p.capm, p.nodes_lds, T / 64, 0u, 0).total + kHubCap * 32;
        const size_t lim2 = (size_t)(160 * 1024) / 2 - 64;
        const size_t lim = base + 128 * per_run <= lim2 ? lim2 : (size_t)160 * 1024 - 256;
        if (base + 128 * per_run > lim) return set_error(SG_ERR_INVALID, "sg_sample: scan LDS layout %zu B too large", base);
        p.run_cap = (uint32_t)std::min<size_t>(1024, (lim - base) / per_run) & ~31u;
        p.run_cap = std::max<uint32_t>(128, std::min<uint32_t>(p.run_cap, env_u32("SHADOW_SG_RUNCAP", 1024)));   // (>= 128: see the kernel)
      }
      const ScanLayout SL = flat ? scan_layout(p.bit_words, p.capm, p.nodes_lds, T / 64, 0u, p.run_cap, p.include_self != 0)
                                  : scan_layout(p.bit_words, p.capm, p.nodes_lds, T / 64);
      if (SL.total > 160 * 1024 - 256) return set_error(SG_ERR_INVALID, "sg_sample: scan LDS layout %zu B too large", SL.total);
      uint32_t per_cu = (uint32_t)std::min<size_t>((size_t)(160 * 1024) / (SL.total + 64), 32 / (T / 64));
      per_cu = std::max<uint32_t>(1, std::min<uint32_t>(per_cu, 8));
      p.scan_grid = std::min<uint32_t>((uint32_t)ncu * per_cu, kScanGridMax);
      hipLaunchKernelGGL(sg_plan_kernel, dim3(1), dim3(1024), 0, stream, p);
      SHD_HIP(hipGetLastError());
      const void *kfn = flat ? (p.include_self ? (const void *)sg_scan_plain_kernel<true> : (const void *)sg_scan_plain_kernel<false>)
                             : plain ? (const void *)sg_scan_kernel<true> : (const void *)sg_scan_kernel<false>;
      if (SL.total > 64 * 1024)
        SHD_HIP(ensure_dynamic_lds(kfn, SL.total));
      if (flat && p.include_self) {
        if (!p.s_selfpos) return set_error(SG_ERR_STATE, "sg_sample: the rows' self-edge slots were not prepared");
        hipLaunchKernelGGL(sg_scan_plain_kernel<true>, dim3(p.scan_grid), dim3(T), SL.total, stream, p);
      }
      else if (flat) hipLaunchKernelGGL(sg_scan_plain_kernel<false>, dim3(p.scan_grid), dim3(T), SL.total, stream, p);
      else if (plain) hipLaunchKernelGGL(sg_scan_kernel<true>, dim3(p.scan_grid), dim3(T), SL.total, stream, p);
      else hipLaunchKernelGGL(sg_scan_kernel<false>, dim3(p.scan_grid), dim3(T), SL.total, stream, p);
      SHD_HIP(hipGetLastError());
    }
    if (s->profiling) SHD_HIP(hipEventRecord(s->ev_t[1], stream));
    RelocParams r;
    memset(&r, 0, sizeof(r));
    r.P = P; r.R = R; r.aug_flags = cfg->aug_flags;
    r.cap_nodes_scr = capn; r.cap_edges_scr = cape;
    r.s_nodes = p.s_nodes; r.s_ppr = p.s_ppr; r.s_rowptr = (uint32_t *)(sc + o_rowptr);
    r.s_row = p.s_row; r.s_col = p.s_col;
    r.s_eid = p.s_eid; r.s_tgt = p.s_tgt; r.s_cnt = p.s_cnt; r.s_tmp = (uint32_t *)(sc + o_tmp);
    r.cstart = p.cstart; r.recs = p.recs; r.blkinfo = p.blkinfo; r.plan = p.plan; r.s_lcol = (uint32_t *)(sc + o_lcol);
    r.nbatch = nbatch; r.bstart[0] = 0;
    for (uint32_t b = 0; b < nbatch; b++) { r.bstart[b + 1] = r.bstart[b] + bsize[b]; r.outs[b] = outs[b]; }
    r.d_counts = s->d_counts;
    // one workgroup per subgraph: subgraphs of thousands of nodes (depth-3 k-hop) get 1024 threads for their edge walk and BFS
    // (measured, 512 roots of the depth-3 benchmark: 0.27 ms at 256 threads)
    const uint32_t reloc_threads = capn >= 2048 ? 1024u : 256u;
    hipLaunchKernelGGL(sg_relocate_kernel, dim3(P), dim3(reloc_threads), 0, stream, r);
    SHD_HIP(hipGetLastError());
    if (s->profiling) SHD_HIP(hipEventRecord(s->ev_t[2], stream));
    s->timed = s->profiling;
  }
  SHD_HIP(hipMemcpyAsync(s->h_counts, s->d_counts, 8 * nbatch * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  SHD_HIP(hipEventRecord(s->ev, stream));
  s->pending = true;
  s->pending_P = P;
  s->pending_nbatch = nbatch;
  for (uint32_t b = 0; b < nbatch; b++) s->pending_bsize[b] = bsize[b];
  return SG_OK;
}

extern "C" int sg_sample(sg_sampler *s, const sg_config *cfg, uint64_t root_start,
                         uint32_t num_subgraphs, uint64_t serial_base,
                         const uint32_t *d_roots_override, const sg_batch_out *out, void *stream_) {
  return sample_impl(s, cfg, root_start, 1, &num_subgraphs, serial_base, d_roots_override, out, stream_);
}

extern "C" int sg_sample_multi(sg_sampler *s, const sg_config *cfg, uint64_t root_start, uint32_t num_batches,
                               const uint32_t *batch_subgraphs, uint64_t serial_base, const uint32_t *d_roots_override,
                               const sg_batch_out *outs, void *stream_) {
  return sample_impl(s, cfg, root_start, num_batches, batch_subgraphs, serial_base, d_roots_override, outs, stream_);
}

extern "C" int sg_sample_finish_multi(sg_sampler *s, uint32_t num_batches, sg_batch_counts *counts) {
  if (!s || !counts) return set_error(SG_ERR_INVALID, "sg_sample_finish: null argument");
  if (!s->pending) return set_error(SG_ERR_STATE, "sg_sample_finish: no sample in flight");
  if (num_batches != s->pending_nbatch)
    return set_error(SG_ERR_INVALID, "sg_sample_finish: the call in flight has %u batches, not %u", s->pending_nbatch, num_batches);
  SHD_HIP(hipSetDevice(s->device));
  SHD_HIP(hipEventSynchronize(s->ev));
  s->pending = false;
  uint32_t overflow = 0, worst = 0;
  for (uint32_t b = 0; b < num_batches; b++) {
    const uint64_t *h = s->h_counts + 8 * b;
    sg_batch_counts *c = counts + b;
    c->n_tot = h[0]; c->e_tot = h[1];
    c->num_subgraphs = s->pending_bsize[b];
    c->max_subg_nodes = (uint32_t)h[2]; c->max_subg_edges = (uint32_t)h[3];
    c->overflow = (uint32_t)h[4];
    c->slots_scanned = h[5]; c->frontier_reads = h[7]; c->frontier_nodes = h[6];
    c->sample_kernel_ms = 0.f; c->relocate_kernel_ms = 0.f;
    if (c->overflow && !overflow) worst = b;
    overflow |= c->overflow;
  }
  // (the kernels of a call serve all of its batches: their durations are reported on the FIRST batch's counts)
  if (s->timed) {
    (void)hipEventElapsedTime(&counts[0].sample_kernel_ms, s->ev_t[0], s->ev_t[1]);
    (void)hipEventElapsedTime(&counts[0].relocate_kernel_ms, s->ev_t[1], s->ev_t[2]);
  }
  // flag 16 (the scan's pool of round records ran out: subgraphs cut into unusually many rounds) is repaired here -- the
  // next call of this sampler carves a pool twice as large -- so that re-issuing the same call converges
  if ((overflow & 16u) && s->rec_scale < 64u) s->rec_scale *= 2u;
  if (overflow)
    return set_error(SG_ERR_CAPACITY,
                     "sg_sample: capacity exceeded (flags=0x%x: 1=subgraph nodes 2=subgraph edges 4=out nodes 8=out edges "
                     "16=round records [pool doubled for the next call]); "
                     "batch %u: largest subgraph %u nodes / %u edges, batch %llu nodes / %llu edges",
                     overflow, worst, counts[worst].max_subg_nodes, counts[worst].max_subg_edges,
                     (unsigned long long)counts[worst].n_tot, (unsigned long long)counts[worst].e_tot);
  return SG_OK;
}

extern "C" int sg_sample_finish(sg_sampler *s, sg_batch_counts *counts) {
  if (s && s->pending && s->pending_nbatch != 1)
    return set_error(SG_ERR_STATE, "sg_sample_finish: a %u-batch call is in flight (sg_sample_finish_multi)", s->pending_nbatch);
  return sg_sample_finish_multi(s, 1, counts);
}

extern "C" int sg_set_profiling(sg_sampler *s, int enable) {
  if (!s) return set_error(SG_ERR_INVALID, "sg_set_profiling: null sampler");
  SHD_HIP(hipSetDevice(s->device));
  if (enable && !s->ev_t[0])
    for (int i = 0; i < 3; i++) SHD_HIP(hipEventCreate(&s->ev_t[i]));
  s->profiling = enable != 0;
  return SG_OK;
}

// Debug / calibration: stream the full-graph rows of `n` nodes (device array) exactly like the scan does -- aligned
// 16-byte quads, one wavefront per row, `depth` 1-KiB loads in flight -- and fold them into one word per row.  No
// filter, no LDS: the rate this reaches is the memory system's ceiling for the sampler's access pattern.
namespace shadow {
template <int kDepth>
__global__ void __launch_bounds__(256) stream_rows_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                          const uint32_t *__restrict__ nodes, uint32_t n, uint32_t *__restrict__ out) {
  const uint32_t lane = lane_id();
  const uint32_t wpb = blockDim.x >> 6;
  for (uint32_t r = blockIdx.x * wpb + wave_id(); r < n; r += gridDim.x * wpb) {
    const uint32_t v = nodes[r];
    const uint32_t e0 = indptr[v], e1 = indptr[v + 1];
    const uint32_t q0 = e0 >> 2, q1 = e1 > e0 ? ((e1 - 1u) >> 2) + 1u : q0;
    uint32_t acc = 0;
    for (uint32_t q = q0; q < q1; q += 64u * kDepth) {
      uint4 c[kDepth];
#pragma unroll
      for (int u = 0; u < kDepth; u++) {
        const uint32_t qq = q + 64u * u + lane;
        c[u] = qq < q1 ? *reinterpret_cast<const uint4 *>(indices + 4ull * qq) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < kDepth; u++) acc ^= c[u].x ^ c[u].y ^ c[u].z ^ c[u].w;
    }
    acc = wave_reduce_sum(acc);
    if (lane == 0) out[r] = acc;
  }
}
}  // namespace shadow

extern "C" int sg_debug_stream_rows(sg_sampler *s, const uint32_t *d_nodes, uint32_t n, uint32_t *d_out, int depth, int blocks,
                                    void *stream) {
  if (!s || !d_nodes || !d_out) return set_error(SG_ERR_INVALID, "sg_debug_stream_rows: null argument");
  SHD_HIP(hipSetDevice(s->device));
  const dim3 grid((unsigned)std::max(1, blocks)), block(256);
  if (depth >= 4) hipLaunchKernelGGL(stream_rows_kernel<4>, grid, block, 0, (hipStream_t)stream, s->d_indptr, s->d_indices, d_nodes, n, d_out);
  else if (depth >= 2) hipLaunchKernelGGL(stream_rows_kernel<2>, grid, block, 0, (hipStream_t)stream, s->d_indptr, s->d_indices, d_nodes, n, d_out);
  else hipLaunchKernelGGL(stream_rows_kernel<1>, grid, block, 0, (hipStream_t)stream, s->d_indptr, s->d_indices, d_nodes, n, d_out);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

extern "C" int sg_debug_scan_phases(sg_sampler *s, uint32_t *h_out16) {
  if (!s || !h_out16) return set_error(SG_ERR_INVALID, "sg_debug_scan_phases: null argument");
  if (!s->last_plan) return set_error(SG_ERR_STATE, "sg_debug_scan_phases: nothing sampled yet");
  SHD_HIP(hipSetDevice(s->device));
  SHD_HIP(hipDeviceSynchronize());
  SHD_HIP(hipMemcpy(h_out16, s->last_plan, PL_WORDS * 4, hipMemcpyDeviceToHost));
  return SG_OK;
}

extern "C" int sg_debug_subgraph_stats(sg_sampler *s, uint32_t *h_out, uint32_t max_subgraphs) {
  if (!s || !h_out) return set_error(SG_ERR_INVALID, "sg_debug_subgraph_stats: null argument");
  if (!s->last_cnt) return set_error(SG_ERR_STATE, "sg_debug_subgraph_stats: nothing sampled yet");
  SHD_HIP(hipSetDevice(s->device));
  SHD_HIP(hipDeviceSynchronize());
  const uint32_t P = std::min(max_subgraphs, s->pending_P);
  SHD_HIP(hipMemcpy(h_out, s->last_cnt, (size_t)P * R_WORDS * 4, hipMemcpyDeviceToHost));
  return SG_OK;
}
