/*
 * shadow_hip.h -- C ABI of libshadow_hip.so: the MI355X (gfx950) native
 * replacement for the shaDow-GNN sampler backend and aggregation ops.
 *
 * Boundary rules: extern "C", plain pointers and sizes, integer status codes,
 * no exceptions, no torch types.  `stream` arguments are hipStream_t passed as
 * void* (NULL = the default stream).  Unless stated otherwise every pointer
 * named d_* is a DEVICE pointer and every pointer named h_* is a HOST pointer.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   para_graph_sampler/graph_engine/backend/ParallelSampler.cpp:707-746
 *       pybind11 module `ParallelSampler` (classes ParallelSampler,
 *       SubgraphStructVec) -- the sg_* entry points below;
 *   shaDow/layers.py:326-327,433,475,580  torch.sparse.mm call sites,
 *   shaDow/layers.py:329-338 (_f_norm_feat), :560-582 (GAT attention),
 *   para_graph_sampler/graph_engine/frontend/graph_utils.py:63-145
 *       (adjacency normalisation) -- the sl_* entry points below.
 */
#ifndef SHADOW_HIP_H
#define SHADOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status */
#define SG_OK 0
#define SG_ERR_INVALID 1  /* bad argument / unsupported configuration           */
#define SG_ERR_HIP 2      /* a HIP runtime call failed (see sg_last_error)      */
#define SG_ERR_CAPACITY 3 /* a scratch/output capacity was too small; counts say
                             how much is needed -- grow and call again          */
#define SG_ERR_IO 4       /* file could not be read / has the wrong format      */
#define SG_ERR_STATE 5    /* call sequence error (e.g. no targets set)          */

/* Last error message of the calling thread ("" if none). Never NULL. */
const char *sg_last_error(void);
/* Library ABI version (bumped on incompatible changes). */
int sg_abi_version(void);

/* ---------------------------------------------------------------- sampler
 * Replaces class ParallelSampler (ParallelSampler.h:25-156).               */
typedef struct sg_sampler sg_sampler;

#define SG_METHOD_KHOP 0    /* ParallelSampler::khop            .cpp:510-556 */
#define SG_METHOD_PPR 1     /* ParallelSampler::ppr             .cpp:565-595 */
#define SG_METHOD_NODEIID 2 /* ParallelSampler::nodeIID         .cpp:498-508 */

#define SG_AUG_HOPS 1  /* SubgraphStruct::compute_hops          Graph.cpp:32-64 */
#define SG_AUG_PPRS 2  /* ppr score per node                    .cpp:365        */
#define SG_AUG_DRNLS 4 /* compute_drnl_single                   Graph.cpp:66-73 */

/* Sampler configuration == the string->string dict the reference frontend
 * hands to C++ (frontend/samplers_cpp.py:40-45,73-80,124-131), parsed.       */
typedef struct sg_config {
  int32_t method;              /* SG_METHOD_*                                  */
  int32_t num_roots;           /* "num_roots": 1 (node task) or 2 (link task)  */
  int32_t depth;               /* khop "depth"                                 */
  int32_t budget;              /* khop "budget" (<0: all neighbours)           */
  int32_t k;                   /* ppr "k"                                      */
  float threshold;             /* ppr "threshold"                              */
  int32_t add_self_edge;       /* "add_self_edge"                              */
  int32_t include_target_conn; /* "include_target_conn"                        */
  int32_t compat_overread;     /* 1: reproduce the reference's unsigned
                                  idx_insert>=0 over-read (.cpp:385,401)       */
  int32_t aug_flags;           /* SG_AUG_* bit set                             */
} sg_config;

/* Caller-provided (e.g. torch-allocated) DEVICE output buffers for one batch,
 * written in block-diagonal form (== frontend/graph.py:280-320 applied to the
 * per-subgraph vectors of SubgraphStructVec, Graph.h:59-97).                 */
typedef struct sg_batch_out {
  uint32_t *d_node;       /* [cap_nodes]   original node id (origNodeID)       */
  uint32_t *d_indptr;     /* [cap_nodes+1] batch-level CSR row pointers        */
  uint32_t *d_indices;    /* [cap_edges]   batch-level column ids              */
  uint32_t *d_edge_id;    /* [cap_edges]   origEdgeID, 0xFFFFFFFF = self edge  */
  uint32_t *d_target;     /* [P*num_roots] batch-level id of each root         */
  uint32_t *d_subg_nodes; /* [P+1] node offset of each subgraph (exclusive scan, last = n_tot) */
  uint32_t *d_subg_edges; /* [P+1] edge offset of each subgraph                */
  uint32_t *d_hop;        /* [cap_nodes] or NULL (SG_AUG_HOPS)                 */
  float *d_ppr;           /* [cap_nodes] or NULL                               */
  uint32_t *d_drnl;       /* [cap_nodes] or NULL (SG_AUG_DRNLS)                */
  uint64_t cap_nodes;
  uint64_t cap_edges;
} sg_batch_out;

typedef struct sg_batch_counts {
  uint64_t n_tot;           /* nodes of the batch                               */
  uint64_t e_tot;           /* edges of the batch                               */
  uint32_t num_subgraphs;
  uint32_t max_subg_nodes;  /* largest subgraph (nodes / edges)                 */
  uint32_t max_subg_edges;
  uint32_t overflow;        /* bit0: per-subgraph node cap, bit1: per-subgraph
                               edge cap, bit2: output cap_nodes, bit3: output
                               cap_edges                                        */
  uint64_t slots_scanned;   /* full-graph neighbour ids examined (= D + n)      */
  uint64_t frontier_reads;  /* neighbour ids read by the k-hop expansion        */
  uint64_t frontier_nodes;  /* frontier nodes expanded (2 indptr reads each)    */
  float sample_kernel_ms;   /* duration of the sampling kernel(s) of this call,
                               HIP events on the launch stream; 0 unless
                               sg_set_profiling(s, 1)                           */
  float relocate_kernel_ms; /* same for the relocation kernel                   */
} sg_batch_counts;

/* Create a sampler over a full-graph CSR (uint32 indptr[N+1], indices[nnz]).
 * on_device = 0: host arrays, uploaded (pinned, chunked) into HBM and owned
 * by the handle; 1: device arrays borrowed from the caller.
 * seed < 0 -> time based (ParallelSampler.h:49-53).
 * Replaces ParallelSampler::ParallelSampler (ParallelSampler.h:27-69).       */
int sg_create(const uint32_t *indptr, const uint32_t *indices, uint32_t num_nodes,
              uint64_t num_edges, int on_device, int device_id, int64_t seed, sg_sampler **out);
/* Same from the reference's raw little-endian uint32 .bin files
 * (ParallelSampler::read_array_from_bin, .cpp:70-86).                        */
int sg_create_from_bin(const char *path_indptr, const char *path_indices, int device_id,
                       int64_t seed, sg_sampler **out);
/* Same with the files' element widths stated (4 = the reference's uint32; 8 = uint64 / int64 as numpy / scipy
 * hold index arrays of graphs with more than 2^31 entries, frontend/loader.py:63-96).  Both files stream
 * through a double-buffered pinned staging area into HBM -- the whole array is never held on the host
 * (papers100M: 13.4 GB).  64-bit elements are narrowed with a range check: a graph with >= 2^32 nodes or
 * edges does not fit the sampler's uint32 ids (Graph.h:16) and is refused with SG_ERR_INVALID.             */
int sg_create_from_bin_ex(const char *path_indptr, const char *path_indices, int indptr_bytes, int indices_bytes,
                          int device_id, int64_t seed, sg_sampler **out);
void sg_destroy(sg_sampler *s);

uint32_t sg_num_nodes(const sg_sampler *s);        /* .cpp:49 */
uint64_t sg_num_edges(const sg_sampler *s);        /* .cpp:51 */
uint64_t sg_num_nodes_target(const sg_sampler *s); /* .cpp:53 */
uint64_t sg_get_idx_root(const sg_sampler *s);     /* .cpp:45 */
const uint32_t *sg_device_indptr(const sg_sampler *s);
const uint32_t *sg_device_indices(const sg_sampler *s);

/* Epoch root order (ParallelSampler::shuffle_targets, .cpp:36-43): h_targets
 * is the pre-shuffled list; it is uploaded once.  Resets nothing else.       */
int sg_shuffle_targets(sg_sampler *s, const uint32_t *h_targets, uint64_t count);
/* Sequential root cursor (ParallelSampler::_get_roots_p, .cpp:456-468):
 * reserves the next min(max_subgraphs, remaining/num_roots) root groups,
 * wraps the cursor to 0 at the end of the list, and reserves as many RNG
 * serial numbers.  Returns the position of the first root and the first
 * serial.                                                                   */
int sg_next_roots(sg_sampler *s, uint32_t num_roots, uint32_t max_subgraphs, uint64_t *root_start,
                  uint32_t *num_subgraphs, uint64_t *serial_base);

/* Per-subgraph scratch capacities (nodes / edges).  0 = derive from the
 * config.  Capacities only ever grow.                                        */
int sg_set_caps(sg_sampler *s, uint32_t cap_subg_nodes, uint32_t cap_subg_edges);
int sg_get_caps(const sg_sampler *s, const sg_config *cfg, uint32_t *cap_subg_nodes,
                uint32_t *cap_subg_edges);

/* PPR table, compacted by root (the reference keeps a dense N-row table,
 * ParallelSampler.h:141-142, only filled for the preprocessed targets).
 * Host arrays: h_targets[R], h_len[R], h_neigh[R*stride], h_score[R*stride]. */
int sg_set_ppr(sg_sampler *s, const uint32_t *h_targets, uint32_t num_rows, const uint32_t *h_len,
               const uint32_t *h_neigh, const float *h_score, uint32_t stride);
/* Read / write the reference's PPR cache files (.cpp:94-231).  alpha is the
 * user-facing value (the files store 1-alpha).  Load returns SG_ERR_IO when
 * the header does not match, exactly as the reference falls back (.cpp:166). */
int sg_load_ppr_bin(sg_sampler *s, const char *path_neighs, const char *path_scores, int k,
                    float alpha, float epsilon);
int sg_save_ppr_bin(const sg_sampler *s, const char *path_neighs, const char *path_scores, int k,
                    float alpha, float epsilon);
/* Approximate PPR by lazy-walk push, one wavefront per target, bit-exact with
 * ParallelSampler::preproc_ppr_approximate (.cpp:237-318): for every target the
 * touched set {(node, pi)} is appended to out_node / out_score at
 * out_offset[t] (out_count[t] entries, unordered); the caller orders them by
 * (-score, id) and keeps k (.cpp:320-339).  `alpha` is the user-facing value
 * (flipped to 1-alpha inside, .cpp:242).  hash_slots = per-target table size
 * (power of two), num_waves = targets in flight (multiple of 4), d_work >=
 * num_waves*hash_slots*21 + 256 bytes.  Synchronises the stream; SG_ERR_CAPACITY
 * when a table or the output list is too small (*h_total = entries needed).
 * mode 0 = "ordered": smallest pending id first, as the reference's std::set (bit-exact tables);
 * mode 1 = "fifo": same push arithmetic in discovery order (ring queue filled by all lanes), several
 * times faster; every residue still ends <= epsilon * degree, scores agree within that bound.      */
int sg_ppr_push(const uint32_t *d_indptr, const uint32_t *d_indices, uint32_t num_nodes,
                const uint32_t *d_targets, uint32_t num_targets, float alpha, float epsilon,
                uint32_t hash_slots, uint32_t num_waves, void *d_work, uint64_t work_bytes,
                uint32_t *d_out_count, uint64_t *d_out_offset, uint32_t *d_out_node, float *d_out_score,
                uint64_t cap_out, uint64_t *h_total, uint32_t *h_flags, uint32_t mode, void *stream);
/* ParallelSampler::drop_full_graph_info (.cpp:22-34). */
int sg_drop_full_graph_info(sg_sampler *s);

/* Sample `num_subgraphs` subgraphs for roots targets[root_start ...) and write
 * the batch into `out`.  Asynchronous on `stream`; call sg_sample_finish to
 * wait and fetch the counts.  Replaces one ensemble branch of
 * ParallelSampler::parallel_sampler_ensemble (.cpp:662-704) plus the Python
 * collate (frontend/graph.py:280-320).
 * d_roots_override: optional DEVICE pointer to num_subgraphs*num_roots roots
 * used instead of the shuffled target list (root_start ignored).            */
int sg_sample(sg_sampler *s, const sg_config *cfg, uint64_t root_start, uint32_t num_subgraphs,
              uint64_t serial_base, const uint32_t *d_roots_override, const sg_batch_out *out,
              void *stream);
int sg_sample_finish(sg_sampler *s, sg_batch_counts *counts);
/* Several consecutive batches from ONE call: the subgraphs of batch b are the next batch_subgraphs[b] root groups behind
 * those of batch b - 1 (roots targets[root_start ...) or the override list, RNG serials serial_base ...), written
 * block-diagonally into outs[b] with offsets relative to that batch -- exactly what num_batches consecutive sg_sample
 * calls write (the draws are keyed on the subgraph's serial number, not on the call), but the four dependent kernels of
 * the pipeline are launched once: at the benchmark's 1 024 roots per step their fixed latency is a third of a call
 * (DESIGN section 3).  The reference has no counterpart: its trainer asks for one batch at a time
 * (ParallelSampler::parallel_sampler_ensemble, .cpp:662-704); a prefetching caller asks for the next few steps' batches.
 * 1 <= num_batches <= SG_MAX_BATCHES_PER_CALL, every batch non-empty.  sg_sample_finish_multi fills counts[num_batches];
 * the kernel durations of the call (profiling on) are reported on counts[0].  A capacity overflow of any batch fails the
 * whole call (SG_ERR_CAPACITY, flags OR-ed): grow and call again. */
#define SG_MAX_BATCHES_PER_CALL 16
int sg_sample_multi(sg_sampler *s, const sg_config *cfg, uint64_t root_start, uint32_t num_batches,
                    const uint32_t *batch_subgraphs, uint64_t serial_base, const uint32_t *d_roots_override,
                    const sg_batch_out *outs, void *stream);
int sg_sample_finish_multi(sg_sampler *s, uint32_t num_batches, sg_batch_counts *counts);
/* Bracket the kernels of every sg_sample call with timing events (bench.py's
 * live roofline measurement).                                                 */
int sg_set_profiling(sg_sampler *s, int enable);

/* ------------------------------------------------------------ layer ops
 * Aggregation / normalisation kernels over a batch CSR (uint32 indptr[n+1],
 * indices[e], block diagonal as produced by sg_sample).  All pointers are
 * DEVICE pointers; features are row-major fp32 with a leading dimension in
 * elements.  These replace the torch.sparse.mm / torch_scatter / scipy call
 * sites of shaDow/layers.py and frontend/graph_utils.py listed per function.  */

/* out[i,:] = table[idx[i],:]      (feat_full[subgs.node], shaDow/minibatch.py:469) */
/* (ABI 24) Up to SL_ROWS_MAX_JOBS row copies under one int64 row index in ONE launch, for i < rows:
 *   mode 0 (gather)   dst[i, :] = src[idx[i], :]       mode 1 (clear)   dst[i, :] = 0       mode 2 (scatter)   dst[idx[i], :] = src[i, :]
 * (scatter: distinct idx; a clear of the same dst belongs in an EARLIER call).  Pure copies: what `x.index_select(0, idx)`,
 * `torch.zeros` and `index_copy_` do around the row-sparse backward passes (ops_gat._GatTail._rows_backward), in two launches
 * instead of fifteen. */
#define SL_ROWS_MAX_JOBS 12
typedef struct {
  const float *src; int64_t lds;   /* source rows and their pitch in floats (unused for mode 1) */
  float *dst; int64_t ldd;
  uint32_t width;                  /* floats per row */
  int mode;
} sl_rows_job;
int sl_rows_multi(const sl_rows_job *jobs, int njobs, const int64_t *d_idx, uint32_t rows, void *stream);

int sl_gather_rows_f32(const float *d_table, int64_t ld_table, const uint32_t *d_idx, uint32_t n,
                       uint32_t F, float *d_out, int64_t ld_out, void *stream);
/* The same gather with layer 0's input dropout in the same pass (nn.Dropout at layers.py:430,471; the counter-hash
 * mask rule of sl_act_norm_fwd with (drop_p, drop_seed), row = batch row i), written into rows padded with zeros up
 * to F_pad columns (whole 128-byte lines: F = 100 -> F_pad = 128), ld_out >= F_pad.  F % 4 == 0, aligned rows.
 * d_row_amax (may be NULL): receives max_k |out[i, k]| per row -- the operand scale of the GEMM that reads the rows
 * next (see sl_row_amax).                                                                                          */
int sl_gather_rows_drop_f32(const float *d_table, int64_t ld_table, const uint32_t *d_idx, uint32_t n, uint32_t F,
                            float drop_p, uint64_t drop_seed, float *d_out, int64_t ld_out, uint32_t F_pad,
                            float *d_row_amax, void *stream);

/* edge_row[p] = row of edge p (the COO row index adj._indices()[0] of
 * frontend/graph_utils.py:48-56, without the host round trip).               */
int sl_csr_edge_rows(const uint32_t *d_indptr, uint32_t n, uint32_t e, uint32_t *d_edge_row,
                     void *stream);

/* Transposed CSR for the backward pass (dX = A^T dY): t_indptr[n+1],
 * t_indices[e] (source rows, ascending) and t_perm[e] (position of the same
 * edge in the original CSR).  d_work: uint32[n + 2*e + 16] scratch.           */
int sl_csr_transpose(const uint32_t *d_indptr, const uint32_t *d_indices, const uint32_t *d_edge_row,
                     uint32_t n, uint32_t e, uint32_t *d_t_indptr, uint32_t *d_t_indices,
                     uint32_t *d_t_perm, uint32_t *d_work, void *stream);

/* Degree scales of the adjacency normalisations (edge_w = drop-edge mask or
 * NULL for all ones):  mode 0 "rw":  1/max(1,sum_j w_ij)     (adj_norm_rw,
 * frontend/graph_utils.py:84-94);  mode 1 "sym": max(1,sum_j w_ij)^-1/2
 * (adj_norm_sym, frontend/graph_utils.py:140-142).                            */
int sl_degree_scales(const uint32_t *d_indptr, const float *d_edge_w, uint32_t n, int mode,
                     float *d_row_scale, void *stream);

/* ---------------------------------------------------------------------------
 * Device-side subgraph cache: the reference's record -> reuse loop for deterministic samplers
 * (CachedSubgraph / PoolSubgraph, shaDow/minibatch.py:21-91; par_graph_sample :403-426;
 * REUSABLE_SAMPLER = {ppr}, CONFIG_TEMPLATE.yml:16-17).  Epoch 1 records every sampled
 * single-root subgraph under its root id; later epochs rebuild any batch of roots from the arena
 * in block-diagonal form (cat_to_block_diagonal, frontend/graph.py:280-330) without sampling --
 * the full graph may then be dropped (sg_drop_full_graph_info), as the reference does at
 * optm_level 'high' (minibatch.py:336-341).
 * ------------------------------------------------------------------------- */
typedef struct sg_cache sg_cache;
int sg_cache_create(uint32_t num_nodes, int device_id, sg_cache **out);
void sg_cache_destroy(sg_cache *c);
int sg_cache_clear(sg_cache *c);
int sg_cache_stats(const sg_cache *c, uint64_t *num_recorded, uint64_t *nodes, uint64_t *edges);
/* Append the subgraphs of a finished sg_sample batch (num_roots = 1) whose root is NOT on file yet to the arena (a root
 * recorded before keeps its first copy: nothing is appended for it, so recording the same roots again -- the
 * reference's 'record' mode with percent_per_epoch < 1 revisits most of them epoch after epoch -- does not grow the
 * arena); n_tot / e_tot from sg_batch_counts size the capacity.  Runs on `stream` and waits for it (the counts of what
 * was new come back to the host: sg_cache_stats is exact).                                                        */
int sg_cache_record(sg_cache *c, const sg_batch_out *batch, uint32_t num_subg, uint64_t n_tot, uint64_t e_tot,
                    void *stream);
/* Rebuild the batch of `d_roots[num_subg]` (device array of root ids) into `out` (same layout and
 * capacity contract as sg_sample; d_drnl is not written).  sg_cache_collate_finish waits and returns
 * the sizes; SG_ERR_STATE if a root was never recorded, SG_ERR_CAPACITY if `out` is too small. */
int sg_cache_collate(sg_cache *c, const uint32_t *d_roots, uint32_t num_subg, sg_batch_out *out, void *stream);
int sg_cache_collate_finish(sg_cache *c, sg_batch_counts *counts);

/* Y[i,:] = row_scale[i] * sum_{p in row i} edge_w[edge_perm[p]] * col_scale[col_p] * X[col_p,:]
 * Any of edge_w / edge_perm / row_scale / col_scale may be NULL (= 1 / identity).
 * Replaces torch.sparse.mm(adj_norm, X) (shaDow/layers.py:326-327,433,475,580). */
int sl_spmm_csr_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                    const uint32_t *d_edge_perm, const float *d_row_scale, const float *d_col_scale,
                    const float *d_X, int64_t ldx, float *d_Y, int64_t ldy, uint32_t n, uint32_t F,
                    void *stream);
/* The same with d_row_amax [n] (may be NULL) = max_k |Y[i, k]| per row (the operand scale of the fp16 GEMM that reads Y,
 * see sl_row_amax) -- from the same pass for 128 < F <= 256, one more pass over Y otherwise.  Round 5: the transposed
 * aggregate of a gradient that lives on a few rows runs here over a FILTERED transposed CSR (tail.TopBackwardPlan).     */
int sl_spmm_csr_amax_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                         const uint32_t *d_edge_perm, const float *d_row_scale, const float *d_col_scale,
                         const float *d_X, int64_t ldx, float *d_Y, int64_t ldy, uint32_t n, uint32_t F,
                         float *d_row_amax, void *stream);

/* Consecutive small subgraphs of a collated batch joined into groups of at most cap_rows rows and cap_edges edges (0: the
 * LDS tile of sl_spmm_blockdiag_f32, 384 rows / 1024 edges): d_group_*_off receive [num_subg + 1] offsets -- the groups,
 * then empty groups up to num_subg, so that they can be passed to sl_spmm_blockdiag_f32 in place of the subgraph offsets
 * (with max_subg_nodes = cap_rows) without reading a count back.  A group is still a diagonal block.  num_subg <= 8191. */
int sl_merge_subgraphs(const uint32_t *d_node_off, const uint32_t *d_edge_off, uint32_t num_subg, uint32_t cap_rows,
                       uint32_t cap_edges, uint32_t *d_group_node_off, uint32_t *d_group_edge_off, void *stream);

/* Same product for a BLOCK-DIAGONAL adjacency (a collated minibatch, graph.py:280-330): rows
 * [node_off[s], node_off[s+1]) only reference columns of the same range and own the edges
 * [edge_off[s], edge_off[s+1]) (sg_batch_out.d_subg_nodes / d_subg_edges; the transpose of such a
 * matrix has the same offsets).  Each subgraph's feature tile is staged in LDS once instead of being
 * gathered per edge; subgraphs beyond the LDS tile (384 rows / 1024 edges) gather from HBM.
 * max_subg_nodes (sg_batch_counts) sizes the tile.
 * d_row_amax (may be NULL; vector layout only): joined by an atomic maximum with max_k |Y[i, k]| per row (see
 * sl_row_amax) -- the caller zeroes it, or pre-fills it with the maxima of other columns of the same GEMM operand. */
int sl_spmm_blockdiag_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                          const uint32_t *d_edge_perm, const float *d_row_scale, const float *d_col_scale,
                          const float *d_X, int64_t ldx, float *d_Y, int64_t ldy, uint32_t n, uint32_t F,
                          const uint32_t *d_subg_node_off, const uint32_t *d_subg_edge_off,
                          uint32_t num_subg, uint32_t max_subg_nodes, float *d_row_amax, void *stream);
/* Layer-0 form of the same product: the input matrix is never materialised by a separate pass -- row i of X is
 * table[ids[i]] (the feature gather feat_full[subgs.node], shaDow/minibatch.py:469), optionally with the layer's
 * input dropout applied (nn.Dropout at layers.py:430,471; the counter-hash mask rule of sl_act_norm_fwd below with
 * (drop_p, drop_seed), row = batch row i) -- read ONCE while the subgraph tile is staged in LDS.  d_Xout (optional,
 * [n, F]) receives the gathered (+ dropped) rows for the layer's other consumers (the self Linear of GraphSAGE, the
 * weight gradients).  Y = diag(row_scale) (A o w) diag(col_scale) X as above.  F % 4 == 0, 16-byte aligned rows.   */
/* Y = A . table[ids] on a block-diagonal batch adjacency: the input rows come through a row map (ids[r] = the row of `table` that
 * stands for batch row r; ids[r] == zero_id: row r is all zeros and is not fetched -- the rows a row-sparse gradient does not
 * reach; 0xFFFFFFFF: no such id); d_row_amax (may be NULL) joins the row maxima of Y as sl_spmm_blockdiag_f32 does.
 * F % 4 == 0, 16-byte aligned rows.                                                                                          */
int sl_spmm_blockdiag_rows_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                               const uint32_t *d_edge_perm, const float *d_row_scale, const float *d_col_scale,
                               const float *d_table, int64_t ldt, const uint32_t *d_ids, float *d_Y, int64_t ldy, uint32_t n,
                               uint32_t F, const uint32_t *d_subg_node_off, const uint32_t *d_subg_edge_off, uint32_t num_subg,
                               uint32_t max_subg_nodes, float *d_row_amax, uint32_t zero_id, void *stream);
int sl_spmm_blockdiag_gather_f32(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
                                 const uint32_t *d_edge_perm, const float *d_row_scale, const float *d_col_scale,
                                 const float *d_table, int64_t ldt, const uint32_t *d_ids, float drop_p,
                                 uint64_t drop_seed, float *d_Xout, int64_t ldxo, float *d_Y, int64_t ldy, uint32_t n,
                                 uint32_t F, const uint32_t *d_subg_node_off, const uint32_t *d_subg_edge_off,
                                 uint32_t num_subg, uint32_t max_subg_nodes, void *stream);

/* Weight gradient of nn.Linear with the same split-bf16 arithmetic:  C[N,K] = A[M,N]^T . B[M,K]
 * (A = dZ, B = the layer input; N, K <= 256 and multiples of 4; operands 16-byte aligned, ld % 4 == 0).
 * The rows are cut into sl_gemm_tn_slices(M) slices, one workgroup each; d_partial holds the per-slice
 * products ([slices, N, K] floats) that a second kernel adds in a fixed order (deterministic).
 * d_a_colsum (may be NULL): receives sum_k A[k, j], j < N -- nn.Linear's bias gradient -- from the same pass (d_partial
 * then holds slices * (N K + N) floats). */
uint32_t sl_gemm_tn_slices(uint32_t M);
int sl_gemm_tn_f32(const float *d_A, int64_t lda, const float *d_B, int64_t ldb, float *d_C, uint32_t M, uint32_t N,
                   uint32_t K, float *d_partial, float *d_a_colsum, void *stream);
/* The same product on two fp16 pieces per element (three matrix-core products per tile instead of six): N = K = 256 only,
 * at most 3040 rows per slice (M <= 3040 * sl_gemm_tn_slices(M)).  d_a_amax / d_b_amax [M]: max_k |row| of the operands
 * (upper bounds work too) -- row r of A is scaled by the power of two that puts it at the top of the fp16 range, row r of
 * B by 2^c over that, c one constant per row slice; rows that are zero in either operand contribute nothing.  Measured
 * error against fp64: that of the bf16 form.  Same slices, d_partial ([slices, N, K], + N per slice with d_a_colsum), fixed-order
 * reduction and d_a_colsum (column sums of the unscaled A) as sl_gemm_tn_f32.  SG_ERR_INVALID for shapes it does not take. */
int sl_gemm_tn_f16(const float *d_A, int64_t lda, const float *d_a_amax, const float *d_B, int64_t ldb, const float *d_b_amax,
                   float *d_C, uint32_t M, uint32_t N, uint32_t K, float *d_partial, float *d_a_colsum, void *stream);
/* Two such products against the same B in ONE launch: C1 = A1^T B, C2 = A2^T B (A2 with A1's pitch and row maxima: the two
 * halves of a K-concatenated operand).  The two workgroups of a row slice are dealt to the same XCD and run in step, so B
 * is fetched from HBM once.  d_a1_colsum / d_a2_colsum (both or neither): the column sums of A1 / A2 as in sl_gemm_tn_f32.
 * d_partial: 2 * sl_gemm_tn_slices(M) * (N * K [+ N with column sums]) floats. */
int sl_gemm_tn_f16_pair(const float *d_A1, const float *d_A2, int64_t lda, const float *d_a_amax, const float *d_B, int64_t ldb,
                        const float *d_b_amax, float *d_C1, float *d_C2, uint32_t M, uint32_t N, uint32_t K, float *d_partial,
                        float *d_a1_colsum, float *d_a2_colsum, void *stream);

/* Segment pooling over the rows of each subgraph: out[s,:] = mean | max | sum of X[node_off[s]:node_off[s+1], :]
 * (mode 0 | 1 | 2; an empty subgraph gives zeros).  Replaces F.embedding_bag(arange(n), feat, offsets, mode)
 * in ResPool (shaDow/layers.py:166-183).  d_argmax [num_subg, F] (max mode; may be NULL otherwise) receives
 * the winning row id for the backward pass: dX[i,f] = dout[s,f] / n_s | dout[s,f]*(argmax[s,f]==i) | dout[s,f]. */
int sl_segment_pool_fwd(const float *d_X, int64_t ldx, const uint32_t *d_node_off, uint32_t num_subg, uint32_t F,
                        int mode, float *d_out, int64_t ldo, uint32_t *d_argmax, void *stream);
int sl_segment_pool_bwd(const float *d_dout, int64_t lddo, const uint32_t *d_node_off, uint32_t num_subg,
                        uint32_t F, int mode, const uint32_t *d_argmax, float *d_dX, int64_t lddx, void *stream);
/* (ABI 24) The gradient of (mean | sum pooling of X, X[rows]) -- the read-out of shaDow/layers.py:154-199 -- as a TABLE instead of its
 * [n, F] expansion: sl_pool_grad_rows fills d_index [n] (row i of the batch -> its table row: its subgraph s < num_subg, or
 * num_subg + k for the row of root k), sl_pool_grad_table fills d_table [num_subg + num_roots, F]: row s = dout[s] (/ n_s for the
 * mean: mode 0; 2 = sum), row num_subg + k = that of root k's subgraph + droots[k] (+ every other root entry on the same row).
 * table[index[i]] equals sl_segment_pool_bwd's dX[i] followed by index_add_(rows, droots), bit for bit; d_dout / d_droots may be
 * NULL (no gradient from that side).  Consumer: sl_gemm_an_bwd_plain / sl_sage_below.plain_row. */
int sl_pool_grad_rows(const uint32_t *d_node_off, uint32_t num_subg, const int64_t *d_rows, uint32_t num_roots, uint32_t n,
                      uint32_t *d_index, void *stream);
int sl_pool_grad_table(const float *d_dout, int64_t lddo, const float *d_droots, int64_t lddr, const uint32_t *d_node_off,
                       uint32_t num_subg, const int64_t *d_rows, uint32_t num_roots, uint32_t n, uint32_t F, int mode,
                       float *d_table, int64_t ldt, void *stream);

/* Entity encodings as bit masks of the active one-hot columns (frontend/graph.py:134-172):
 * kind 0 hops (uint32, 0xFFFFFFFF unreachable -> column 0, hop h <= dim-2 -> column h+1, h >= 255 -> column 0),
 * kind 1 pprs (float; column c iff 0.25^c >= ppr >= 0.25^(c+1), last bin down to 0 -- edges belong to both bins),
 * kind 2 drnls (uint32; values >= 255 or > dim-1 -> column 0).  dim <= 32. */
int sl_encode_codes(int kind, const void *d_src, uint32_t n, uint32_t dim, uint32_t *d_codes, void *stream);

/* out = X + onehot(codes) @ W^T + bias without materialising the one-hot matrix (the feature-augmentation
 * Linear of DeepGNN.forward, shaDow/models.py:178-191, 'sum' mode).  d_Wt = W^T, [dim, F] row-major; d_X and
 * d_bias may be NULL; dim <= 16.  Backward: d_dWt [dim, F] and d_dbias [F] (may be NULL) from dout, through
 * d_partial [partial_blocks * (dim+1) * F] (deterministic two-stage sum); dX = dout. */
int sl_onehot_linear_fwd(const float *d_X, int64_t ldx, const uint32_t *d_codes, const float *d_Wt,
                         const float *d_bias, uint32_t n, uint32_t F, uint32_t dim, float *d_out, int64_t ldo,
                         void *stream);
int sl_onehot_linear_bwd(const float *d_dout, int64_t lddo, const uint32_t *d_codes, uint32_t n, uint32_t F,
                         uint32_t dim, float *d_dWt, float *d_dbias, float *d_partial, uint32_t partial_blocks,
                         void *stream);

/* Dense feature x weight product of nn.Linear (shaDow/layers.py:433-435, :474-483, :604-611) on the bf16
 * matrix cores with fp32-level accuracy ("split-bf16": each fp32 operand is split exactly into three bf16
 * pieces and the six largest cross products are accumulated in fp32; error <= 2^-21 |a||b| per product):
 *     C[M,N] = A[M,K] . B[N,K]^T,   fp32 row-major in and out, N <= 256.
 * sl_gemm_pack_b converts B (the Linear weight, [out, in]) once per call into the per-lane MFMA fragment
 * image (sl_gemm_pack_bytes(N, K) bytes); sl_gemm_nt_f32 streams A and writes C.  A must be 16-byte
 * aligned with lda % 4 == 0. */
size_t sl_gemm_pack_bytes(uint32_t N, uint32_t K);
int sl_gemm_pack_b(const float *d_B, int64_t ldb, uint32_t N, uint32_t K, void *d_packed, void *stream);
int sl_gemm_nt_f32(const float *d_A, int64_t lda, const void *d_packed_B, float *d_C, int64_t ldc, uint32_t M,
                   uint32_t N, uint32_t K, void *stream);
/* The same image from strided sources: B element (j, k) = B1[j * s1j + k * s1k] for k < K1 and
 * B2[j * s2j + (k - K1) * s2k] for K1 <= k < K -- a weight packed transposed (s1j = 1, s1k = ld), or the
 * concatenation [Ws^T | Wn^T] of the GraphSAGE input gradient without materialising it.                      */
int sl_gemm_pack_b2(const float *d_B1, int64_t s1j, int64_t s1k, uint32_t K1, const float *d_B2, int64_t s2j, int64_t s2k,
                    uint32_t N, uint32_t K, void *d_packed, void *stream);

/* Fused (bias +) activation + feature normalisation + branch sum:
 *   out = out_scale * sum_{b<nb} ( (h_b - mean) * scale[b] * rsqrt(var + 1e-9) + offset[b] ),
 *   h_b = act_b(Z_b + bias_b), mean/var (biased) over segments of `seg` features
 * (nn.Linear bias + act + shaDowLayer._f_norm_feat, shaDow/layers.py:329-338; GCN
 * :434-435, GraphSAGE :476-483, GAT :620-625 with seg = head slice, out_scale 0.5).
 * act codes: 0 identity("I"), 1 relu, 2 elu, 3 tanh, 4 leakyrelu(0.2).
 * d_Z / ldz / d_bias / act are HOST arrays of nb entries (d_bias or its entries
 * may be NULL); scale / offset are [nb, F].
 * drop_p > 0 fuses the dropout the NEXT layer applies to its input (nn.Dropout at layers.py:430,471,601)
 * into this output: element (row r, column c) is kept iff the 16-bit field (c odd: the high half, c even: the low half) of
 *   mix32(mix32(r_lo ^ seed_lo) + r_hi + seed_hi + (c >> 1) * 0x9E3779B1)                (32-bit wrap-around)
 * is >= clamp(floor(drop_p * 65536), 1, 65535)   (ABI 23; before: one hash per element against drop_p * 2^32 -- the hash's
 * quarter-rate multiplies were 7 % of the forward GEMM-epilogue launch);
 * mix32 = the murmur3 finaliser (h ^= h>>16; h *= 0x85EBCA6B; h ^= h>>13; h *= 0xC2B2AE35; h ^= h>>16);
 * kept values are scaled by 1 / (1 - drop_p); the backward entry regenerates the same mask from
 * (drop_p, drop_seed), no mask tensor exists.  Vector layout only (F % 4 == 0, F <= 256, 16-byte aligned operands).
 * d_out_dropped != NULL (needs drop_p > 0) is the dual mode for architectures where something besides the next
 * layer reads this output (the residue / ResPool read-outs, shaDow/models.py:176-185): d_out receives the plain
 * value and d_out_dropped the dropped one, from one pass over Z.                                                      */
/* 1 when sl_act_norm_* run the vector kernel for (F, seg) given aligned operands (fused dropout, row maxima and the
 * selected-rows backward exist there only).                                                                         */
int sl_act_norm_vector_layout(uint32_t F, uint32_t seg);
/* d_out_amax (may be NULL): receives max_k |row| of the output the next layer reads (the dropped one in dual mode), see
 * sl_row_amax.                                                                                                     */
int sl_act_norm_fwd(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                    const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                    uint32_t seg, float out_scale, float *d_out, int64_t ldo, float drop_p, uint64_t drop_seed,
                    float *d_out_dropped, int64_t ldo_dropped, float *d_out_amax, void *stream);
/* Backward of the above (d_dz0_amax, may be NULL: receives max_k |dZ_0[i, k]| per row, see sl_row_amax; d_row_idx, may be
 * NULL: the output gradient is given for n SELECTED rows only -- d_dout / d_dout_dropped are [n, F] compact and row i of them
 * belongs to row d_row_idx[i] of Z / dZ / the dropout mask: a read-out that takes a few rows of the layer's output; rows of
 * dZ that are not selected are NOT written): dZ_b (entries may be NULL), dscale / doffset [nb, F] and,
 * when d_dbias != NULL, dbias [nb, F] = column sums of dZ_b (all overwritten,
 * reduced over the rows in a fixed order).
 * d_partial: float[2048 * nb * 3 * F] scratch for the two-stage reduction.
 * Dual mode: d_dout_dropped != NULL is the gradient of d_out_dropped (goes through the mask), d_dout the gradient
 * of the plain output and may then be NULL (= zero).                            */
/* sl_act_norm_bwd with d_row_idx and dz_compact != 0: d_dZ[b] / d_dz0_amax are compact too ([n, F] / [n] in the order of
 * d_row_idx) instead of scattered into full-height buffers -- only Z and the dropout mask are addressed through the row ids. */
int sl_act_norm_bwd_rows(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                         const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                         uint32_t seg, float out_scale, const float *d_dout, int64_t lddo, float *const *d_dZ,
                         const int64_t *lddz, float *d_dscale, float *d_doffset, float *d_dbias,
                         float *d_partial, float drop_p, uint64_t drop_seed, const float *d_dout_dropped,
                         int64_t lddo_dropped, float *d_dz0_amax, const uint32_t *d_row_idx, int dz_compact,
                         void *stream);
/* sl_act_norm_bwd_rows that also leaves d_t_out[row, F / seg] = sum over each normalisation segment of dZ[t_branch] * Z[t_branch]
 * for an IDENTITY branch without a bias (vector layout only: sl_act_norm_vector_layout) -- for the GAT layer, whose aggregate N is
 * such a branch, the attention backward's t_i = dN_i . N_i per head (sl_gat_bwd's d_t).  The normalisation is invariant under
 * scaling of its input row, so this dot is S eps rstd^2 m2 (m2 = the segment mean of dy scale xhat the backward pass forms
 * anyway): written from that closed form, no extra reduction.  d_t_out NULL: as without.  amax_branch (ABI 24): the branch whose
 * dZ row maxima d_dz0_amax receives (0 in every other entry) -- the GAT tail asks for its second branch's (dz_self) without
 * re-ordering the [nb, F] scale / offset rows.                                                                                  */
int sl_act_norm_bwd_rows_t(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                           const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                           uint32_t seg, float out_scale, const float *d_dout, int64_t lddo,
                           float *const *d_dZ, const int64_t *lddz, float *d_dscale,
                           float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                           const float *d_dout_dropped, int64_t lddo_dropped, float *d_dz0_amax, const uint32_t *d_row_idx,
                           int dz_compact, int t_branch, float *d_t_out, int amax_branch, void *stream);
/* (ABI 24) sl_act_norm_bwd with the output gradient given as a TABLE: row i's gradient is d_dout[d_dout_map[i], :] -- the gradient of
 * a mean / sum pooling read-out (shaDow/layers.py:166-183) has one row per subgraph (+ one per root), sl_pool_grad_table builds that
 * table and the map; the [n, F] expansion is never written.  Same arithmetic on the same values: bit-identical to sl_act_norm_bwd
 * on the expanded tensor.  d_dout_dropped (dual mode) stays a dense [n, F] tensor.                                                */
int sl_act_norm_bwd_map(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                        const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                        uint32_t seg, float out_scale, const float *d_dout, int64_t lddo, const uint32_t *d_dout_map,
                        float *const *d_dZ, const int64_t *lddz, float *d_dscale, float *d_doffset, float *d_dbias,
                        float *d_partial, float drop_p, uint64_t drop_seed, const float *d_dout_dropped, int64_t lddo_dropped,
                        float *d_dz0_amax, void *stream);
int sl_act_norm_bwd(int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                    const int *act, const float *d_scale, const float *d_offset, uint32_t n, uint32_t F,
                    uint32_t seg, float out_scale, const float *d_dout, int64_t lddo, float *const *d_dZ,
                    const int64_t *lddz, float *d_dscale, float *d_doffset, float *d_dbias,
                    float *d_partial, float drop_p, uint64_t drop_seed, const float *d_dout_dropped,
                    int64_t lddo_dropped, float *d_dz0_amax, const uint32_t *d_row_idx,
                    void *stream);

/* A normalised batch adjacency  diag(row_scale) (A o edge_w) diag(col_scale)  as the layer entries below take it
 * (what ops.NormAdj holds): any of edge_w / row_scale / col_scale may be NULL (= ones); t_* = the transposed CSR
 * with the permutation into the original edge order (sl_csr_transpose), needed by the backward entries when the
 * input gradient is wanted; subg_* = the block offsets of a collated batch (NULL: plain CSR kernels).          */
typedef struct {
  const uint32_t *indptr, *indices;
  const float *edge_w, *row_scale, *col_scale;
  const uint32_t *t_indptr, *t_indices, *t_perm;
  const uint32_t *subg_node_off, *subg_edge_off;
  uint32_t num_subg, max_subg_nodes, n, e;
  uint32_t row_entries_bound;   /* an upper bound of a row's entries when the caller knows one (the k of a top-k PPR batch), 0: unknown.
                                 * More than 64 in a batch of at least 98 304 rows: the aggregations of 256-float rows run on the
                                 * pipelined CSR kernel (sl_set_spmm_wide_pipe)                                                     */
} sl_norm_adj;

/* One GraphSAGE layer pass per call (shaDow/layers.py:471-483 and its autograd):
 *     out = norm_0(act(X Ws^T + bs)) + norm_1(act((A X) Wn^T + bn))      [+ the next layer's input dropout, as
 *     sl_act_norm_fwd does it; d_out_dropped != NULL: dual mode]
 * forward: SpMM -> weight packs -> two split-bf16 GEMMs -> fused bias / act / norm, enqueued by ONE call; the caller
 * provides AX [n, Fin] (ldax), Zs, Zn, out [n, Fout] and d_pack (sl_sage_pack_bytes(n, Fin, Fout) bytes).
 * backward: act_norm backward -> A^T dZn -> dX = [dZs | A^T dZn] . [Ws ; Wn] (one K = 2 Fout product; d_dX NULL: not
 * wanted) -> the two weight gradients; d_buf is [n, 3 Fout] scratch, d_an_partial as for sl_act_norm_bwd (nb = 2),
 * d_tn_partial as for sl_gemm_tn_f32; d_dbias [2, Fout] or NULL.  Fout % 4 == 0, <= 256 (input gradient: Fout % 32
 * == 0, Fin <= 256).  Forward: when sl_gemm_act_norm_supported(Fout, Fin) the two products and the act / norm run as ONE
 * kernel (sl_gemm_act_norm_fwd below), otherwise the same kernels in the same order as the separate entries.         */
/* d_pack: sl_sage_pack_bytes(n, Fin, Fout) bytes of scratch (weight images + the operands' row maxima).
 * d_x_amax (may be NULL): max_k |X[i, k]| per row (sl_row_amax) when the producer of X already wrote it;
 * d_out_amax (may be NULL): receives the row maxima of the dropped output (of `out` without dropout) -- the d_x_amax
 * of the next layer -- from the GEMM epilogue while the rows are in registers.
 * x_pad_zero != 0: the caller states that columns [Fin, roundup32(Fin)) of every row of X are zero (ldx reaches that far;
 * the rows sl_gather_rows_drop_f32 writes are).  With rows of X and A X on whole 128-byte lines the aggregation then
 * zero-fills the same columns of d_AX and the products read both at the padded width (no K tail).                 */
size_t sl_sage_pack_bytes(uint32_t n, uint32_t Fin, uint32_t Fout);
int sl_sage_fwd(const sl_norm_adj *adj, const float *d_X, int64_t ldx, uint32_t Fin, uint32_t Fout, const float *d_Ws,
                int64_t ldws, const float *d_bs, const float *d_Wn, int64_t ldwn, const float *d_bn, const float *d_scale,
                const float *d_offset, int act, float drop_p, uint64_t drop_seed, float *d_AX, int64_t ldax, float *d_Zs,
                float *d_Zn, float *d_out, float *d_out_dropped, const float *d_x_amax, float *d_out_amax,
                void *d_pack, int x_pad_zero, float *d_row_stats, void *stream);
int sl_sage_bwd(const sl_norm_adj *adj, const float *d_X, int64_t ldx, const float *d_AX, int64_t ldax, const float *d_Zs,
                const float *d_Zn, uint32_t Fin, uint32_t Fout, const float *d_Ws, int64_t ldws, const float *d_bs,
                const float *d_Wn, int64_t ldwn, const float *d_bn, const float *d_scale, const float *d_offset, int act,
                float drop_p, uint64_t drop_seed, const float *d_dout, const float *d_dout_dropped, float *d_dX,
                float *d_dWs, float *d_dWn, float *d_dbias, float *d_dscale, float *d_doffset, float *d_buf,
                float *d_an_partial, float *d_tn_partial, void *d_pack, void *stream);

/* GEMM-epilogue forms (csrc/gemm_fused.hip): the activation + feature normalisation of a layer (layers.py:329-338)
 * inside the GEMM that produces (forward) or consumes (backward) its operand -- a wavefront owns 32 whole output rows,
 * the unit `_f_norm_feat` works on, so there is no separate pass over the [n, F] pre-activations.
 *
 * Arithmetic: fp32 operands on the fp16 matrix cores as TWO fp16 pieces per element.  A row of an operand is scaled by a
 * power of two that puts its largest magnitude into [2^14, 2^15) (exact, taken out of the accumulators again) and every
 * element is h + m, h = rn16(x), m = rn16(x - h): 23 significand bits for every element within 2^-17 of its row's
 * maximum, an absolute 2^-39 of that maximum below.  The product keeps hh + hm + mh (the dropped mm <= 2^-22 |ab|,
 * zero-mean): three v_mfma_f32_32x32x16_f16 per tile and k-step, fp32 accumulation -- the measured error against fp64 is
 * that of the six-term bf16 form of sl_gemm_nt_f32 and below a plain fp32 GEMM's (tests: test_split_gemm_matches_fp64).
 * The largest magnitude of every row of A is an input (d_a_amax, [M] floats per operand; the scale follows from it; the
 * array of arrays or single entries may be NULL: the kernel then reads the rows once more itself -- small batches):
 * sl_row_amax computes it in one pass over A; the kernels that produce an operand write it while the row is in their
 * registers (d_out_amax here, d_row_amax of sl_spmm_blockdiag_f32 / sl_gather_rows_drop_f32) and the pass disappears.
 *
 *   sl_gemm_act_norm_fwd   Z_b = A_b . W_b^T for b < nb <= 2 in ONE launch (d_packed_B: sl_gemm_act_norm_pack of the nb
 *                          weights; all branches share M, N, K), Z_b written (without bias), and
 *                          out = out_scale * sum_b norm_b(act_b(Z_b + bias_b)) [+ dropout / dual output exactly as
 *                          sl_act_norm_fwd defines them] written from the accumulators: no second pass over Z.
 *                          N % 4 == 0, 16 <= N <= 256 (normalisation segment = N); operands 16-byte aligned, ld % 4 == 0.
 *                          d_out_amax (may be NULL): receives the row maxima of the dropped output (of out when there
 *                          is no dropout) for the GEMM that reads it next.
 *   sl_gemm_an_bwd         G = A . B^T (A [M, K], d_packed_B = sl_gemm_act_norm_pack_b2, G [M, N]) is the gradient of a
 *                          GraphSAGE layer's output (through that layer's fused output dropout (drop_p, drop_seed) when
 *                          drop_p > 0); G is not written: the epilogue applies sl_act_norm_bwd (nb = 2, seg = N) to it row
 *                          by row and writes dZ_b, dscale, doffset and (d_dbias != NULL) dbias.  d_partial:
 *                          sl_gemm_an_bwd_partial_floats(M, N, nb) floats (one partial row per workgroup, added in a
 *                          fixed order).  d_dz0_amax (may be NULL): receives the row maxima of dZ_0.
 * sl_sage_fwd / sl_gcn_fwd use the forward form whenever sl_gemm_act_norm_supported(Fout, Fin).  sl_set_fused_epilogue(0 | 1)
 * switches it (returns the previous setting; a negative argument only queries; the initial state is ON -- the former
 * environment variable SHADOW_FUSED_EPILOGUE is no longer read) -- the A/B handle of the tests and benchmarks: the layer
 * entries then run sl_gemm_nt_f32 + sl_act_norm_* as separate launches.                                                         */
int sl_set_fused_epilogue(int on);
/* Aggregations of rows wider than 128 floats inside the sl_sage_* / sl_gcn_* entries: the pipelined CSR kernel (a row per
 * wavefront, sl_spmm_csr_amax_f32: its long-row loop keeps six gathers in flight) or the block-diagonal LDS kernel (every edge
 * an LDS read, but a row's entries are walked by ONE thread per float4 column).  1 (default): the CSR kernel for batches whose
 * rows may be long (sl_norm_adj.row_entries_bound > 64 and n >= 98 304: top-k PPR subgraphs, whose root rows list up to k - 1 neighbours), the
 * LDS kernel otherwise; 0: always the LDS kernel; 2: always the CSR kernel; < 0: query.  Returns the previous setting.  Identical
 * sums (both keep the edge order).                                                                                                */
int sl_set_spmm_wide_pipe(int on);
int sl_gemm_act_norm_supported(uint32_t N, uint32_t K);
/* d_amax[i] = max_k |A[i, k]| of a row-major fp32 operand A [n, K] (16-byte aligned, lda % 4 == 0).  The kernels scale
 * row i by the power of two that puts it into [2^14, 2^15) (1 for an all-zero row; exponent clamped to +-62).          */
int sl_row_amax(const float *d_A, int64_t lda, uint32_t n, uint32_t K, float *d_amax, void *stream);
/* Weight images of the epilogue kernels: fp16 pieces in sl_gemm_act_norm_tiles(N) (4 or 8) column tiles, zero above N,
 * followed by a trailer with the power-of-two scale of every weight row and its inverse.  sl_gemm_act_norm_pack writes
 * the nb images of one forward launch (nb * sl_gemm_act_norm_pack_bytes(N, K) bytes: images back to back, then the
 * trailers) in ONE kernel launch that also clears n_zero floats at d_zero (may be NULL: the row-maximum array an SpMM of
 * the same pass joins into); sl_gemm_act_norm_pack_b2 one image from strided / concatenated sources as sl_gemm_pack_b2.             */
uint32_t sl_gemm_act_norm_tiles(uint32_t N);
size_t sl_gemm_act_norm_pack_bytes(uint32_t N, uint32_t K);
int sl_gemm_act_norm_pack(int nb, const float *const *d_B, const int64_t *ldb, uint32_t N, uint32_t K, void *d_packed,
                          float *d_zero, uint32_t n_zero, void *stream);
int sl_gemm_act_norm_pack_b2(const float *d_B1, int64_t s1j, int64_t s1k, uint32_t K1, const float *d_B2, int64_t s2j, int64_t s2k,
                             uint32_t N, uint32_t K, void *d_packed, void *stream);
int sl_gemm_act_norm_fwd(int nb, const float *const *d_A, const int64_t *lda, const float *const *d_a_amax, const void *d_packed_B,
                         uint32_t M, uint32_t N, uint32_t K, float *const *d_Z, const int64_t *ldz, const float *const *d_bias,
                         const int *act, const float *d_scale, const float *d_offset, float out_scale, float *d_out, int64_t ldo,
                         float drop_p, uint64_t drop_seed, float *d_out_dropped, int64_t ldo_dropped, float *d_out_amax,
                         float *d_row_stats, void *stream);
/* d_row_stats (may be NULL; sl_gemm_act_norm_fwd writes, sl_gemm_an_bwd / sl_sage_below.stats read): [M, 2 nb] floats, per row and
 * branch b the pair (mean, 1 / sqrt(var + eps)) of act(Z_b + bias_b) that `_f_norm_feat` (layers.py:334-336) normalises with -- 16
 * bytes per row that spare the backward epilogue two of its four row reductions per branch.                                   */
/* Plain products on the same kernel: C_b = A_b . W_b^T + bias_b for b < nb <= 2 in ONE launch (images:
 * sl_gemm_act_norm_pack / _pack_b2; the two products may share A -- GAT's self and neighbour Linear of the same input;
 * d_bias and its entries may be NULL); d_a_amax as above.  N % 4 == 0, 16 <= N <= 256; operands 16-byte aligned, ld % 4 == 0.
 * sl_gemm_nt_cat_f32: C = [A0 | A1] . B^T + bias with the K-concatenated operand in two tensors (K0 % 32 == 0 columns from
 * A0, K - K0 from A1; image: sl_gemm_act_norm_pack_b2 of the matching weight; d_a_amax: the maximum over BOTH parts of a
 * row, or NULL) -- dX = dZs Ws + dZn Wn of a layer with two Linears on the same input, without the sum pass.           */
int sl_gemm_nt2_f32(int nb, const float *const *d_A, const int64_t *lda, const float *const *d_a_amax, const void *d_packed_B,
                    uint32_t M, uint32_t N, uint32_t K, const float *const *d_bias, float *const *d_C, const int64_t *ldc,
                    void *stream);
/* GAT's two Linears of one input (shaDow/layers.py:604-611) WITH the attention's per-node terms (layers.py:566-569) where the
 * tiles leave the kernel: d_z_self = A W0^T + b0, d_u_s[row, h] = att[0, h] . act(z_self[row, head h]); d_hn = act(A W1^T + b1)
 * (z_neigh itself is not written), d_u_n[row, h] = att[1, h] . hn[row, head h].  d_packed_B: sl_gemm_act_norm_pack of (W0, W1);
 * d_att [2, heads, N / heads]; act as sl_gat_fwd.  Replaces sl_gemm_nt2_f32 + the per-node pass of sl_gat_fwd (then
 * sl_gat_fwd_rows); bit-identical hn / u_s / u_n.  sl_gemm_nt2_gat_supported(N, heads): N == 256, head width 4 * 2^k <= 128. */
int sl_gemm_nt2_gat_supported(uint32_t N, uint32_t heads);
int sl_gemm_nt2_gat_f32(const float *d_A, int64_t lda, const float *d_a_amax, const void *d_packed_B, uint32_t M, uint32_t N, uint32_t K,
                        const float *const *d_bias, float *d_z_self, int64_t ldzs, float *d_hn, int64_t ldhn, const float *d_att, int act,
                        uint32_t heads, float *d_u_s, float *d_u_n, void *stream);
int sl_gemm_nt_cat_f32(const float *d_A0, int64_t lda0, uint32_t K0, const float *d_A1, int64_t lda1, const float *d_a_amax,
                       const void *d_packed_B, uint32_t M, uint32_t N, uint32_t K, const float *d_bias, float *d_C, int64_t ldc,
                       void *stream);
size_t sl_gemm_an_bwd_partial_floats(uint32_t M, uint32_t N, int nb);
int sl_gemm_an_bwd(const float *d_A, int64_t lda, const float *d_a_amax, const void *d_packed_B, uint32_t M, uint32_t N, uint32_t K,
                   int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias, const int *act,
                   const float *d_scale, const float *d_offset, float out_scale, float *const *d_dZ, const int64_t *lddz,
                   float *d_dscale, float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                   float *d_dz0_amax, const float *d_row_stats, void *stream);
/* The same with a ROW-SPARSE ADDEND of the product: G = A . B^T + scatter(corr), row i of G gets d_corr[d_corr_row[i], :]
 * ([corr_rows, N], pitch ldcorr) when d_corr_row[i] < corr_rows.  What it is for: [dZs | A^T dZn] . [Ws ; Wn] with dZs non-zero on
 * a few rows only (the layer below a row-sparse top pass) = (A^T dZn) . Wn-part as a K = Fout product + (dZs[T] . Ws-part) on those
 * rows, added before the act + norm backward of the epilogue uses G.  d_corr NULL: sl_gemm_an_bwd.                              */
int sl_gemm_an_bwd_corr(const float *d_A, int64_t lda, const float *d_a_amax, const void *d_packed_B, uint32_t M, uint32_t N, uint32_t K,
                        int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias, const int *act,
                        const float *d_scale, const float *d_offset, float out_scale, float *const *d_dZ, const int64_t *lddz,
                        float *d_dscale, float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                        float *d_dz0_amax, const float *d_row_stats, const float *d_corr, int64_t ldcorr, const uint32_t *d_corr_row,
                        uint32_t corr_rows, void *stream);
/* ... and with a DENSE addend d_dout_plain [M, N] (pitch lddp; may be NULL: sl_gemm_an_bwd_corr) that does NOT pass the dropout mask:
 * the layer below is in dual-output mode (its plain output feeds a read-out -- residue / pooling, shaDow/models.py:176-185 -- its
 * dropped output this layer), the epilogue forms  dy = G * mask / (1 - p) + d_dout_plain * out_scale  before its act + norm backward:
 * sl_act_norm_bwd's (d_dout, d_dout_dropped) sum.  128 < N <= 256.  d_plain_row (may be NULL): row i's addend is row d_plain_row[i] of
 * d_dout_plain (ABI 24: the gradient table of a mean / sum pooling read-out, sl_pool_grad_table).                                   */
int sl_gemm_an_bwd_plain(const float *d_A, int64_t lda, const float *d_a_amax, const void *d_packed_B, uint32_t M, uint32_t N, uint32_t K,
                         int nb, const float *const *d_Z, const int64_t *ldz, const float *const *d_bias, const int *act,
                         const float *d_scale, const float *d_offset, float out_scale, float *const *d_dZ, const int64_t *lddz,
                         float *d_dscale, float *d_doffset, float *d_dbias, float *d_partial, float drop_p, uint64_t drop_seed,
                         float *d_dz0_amax, const float *d_row_stats, const float *d_corr, int64_t ldcorr, const uint32_t *d_corr_row,
                         uint32_t corr_rows, const float *d_dout_plain, int64_t lddp, const uint32_t *d_plain_row, void *stream);

/* Backward of a GraphSAGE layer chained with the layer below it (consecutive GraphSAGE layers where nothing but this
 * layer reads the lower layer's output -- residue 'none' + centre pooling, shaDow/layers.py:159-163): the input gradient
 * of this layer IS the output gradient of the layer below, so instead of writing it (d_dX) the K = 2 Fout product's
 * epilogue produces the lower layer's dZs / dZn / dscale / doffset / dbias (sl_gemm_an_bwd) into the buffers `below`
 * names; the lower layer's own call then passes dz_ready = 1 (its d_buf = below->buf, its d_dzs_amax = below->amax;
 * d_dout*, d_dscale, d_doffset,
 * d_an_partial and the forward tensors Zs / Zn are not touched) and only runs A^T dZn, its own input-gradient product
 * and the two weight gradients.  d_dout_rows (may be NULL; dz_ready = 0 only): the output gradient is given for num_dout_rows
 * selected rows only (d_dout / d_dout_dropped [num_dout_rows, Fout] compact, see sl_act_norm_bwd) -- the top layer under a
 * read-out that takes the roots' rows.  d_x_amax (may be NULL): max_k |X[i, k]| per row, as the kernel that produced X left
 * it -- with it (and n >= 32768, Fin = Fout = 256) the two weight gradients run on two fp16 pieces (sl_gemm_tn_f16), the
 * neighbour branch as dWn = (A^T dZn)^T X over the transposed aggregate of the input-gradient product (d_AX is then not
 * read, and d_tn_partial must hold TWICE the floats of sl_gemm_tn_f32's: sl_gemm_tn_f16_pair).
 * d_dout_map (ABI 24; may be NULL; dz_ready = 0, no d_dout_rows): row i's d_dout is row d_dout_map[i] of the table d_dout points to
 * (sl_pool_grad_table: a pooled read-out's gradient), sl_act_norm_bwd_map.
 * sl_sage_bwd == sl_sage_bwd_chain(..., 0, NULL, NULL, NULL, 0, NULL, NULL, stream).
 * d_dWs == d_dWn == NULL (round 4): the weight gradients are not computed (a caller that knows dZ to be non-zero on a few rows
 * only forms them on those rows itself); d_tn_partial may then be NULL too.                                                  */
typedef struct {
  const float *Zs, *Zn;            /* [n, F] pre-activations of the layer below (dense rows) */
  const float *bs, *bn;            /* its biases (may be NULL) */
  const float *scale, *offset;     /* [2, F] */
  int act;
  float drop_p;                    /* its fused OUTPUT dropout (= this layer's input dropout), 0: none */
  uint64_t drop_seed;
  uint32_t F;                      /* its output width (= this layer's Fin) */
  float *buf;                      /* [n, 3 F]: receives dZs in columns [0, F) and dZn in [2 F, 3 F) */
  float *dscale, *doffset, *dbias; /* [2, F] each; dbias may be NULL */
  float *partial;                  /* sl_sage_chain_partial_floats(n, F) floats */
  float *amax;                     /* [n]: receives max_k |dZs[i, k]| (the lower layer's call passes it as d_dzs_amax) */
  const float *stats;              /* [n, 4] row statistics its forward pass left (sl_sage_fwd d_row_stats), or NULL */
  const float *dout_plain;         /* (ABI 23) dual-output layer below: the gradient of its PLAIN output, [n, F] dense, added unmasked
                                      before its act + norm backward (sl_gemm_an_bwd_plain); NULL: a single-output layer */
  const uint32_t *plain_row;       /* (ABI 24) NULL, or [n]: row i's plain gradient is row plain_row[i] of dout_plain -- the table of a
                                      pooled read-out's gradient (sl_pool_grad_table) instead of its [n, F] expansion */
  int64_t plain_ld;                /* pitch of dout_plain in floats (0: F) */
} sl_sage_below;
size_t sl_sage_chain_partial_floats(uint32_t n, uint32_t F);
int sl_sage_bwd_chain(const sl_norm_adj *adj, const float *d_X, int64_t ldx, const float *d_AX, int64_t ldax, const float *d_Zs,
                      const float *d_Zn, uint32_t Fin, uint32_t Fout, const float *d_Ws, int64_t ldws, const float *d_bs,
                      const float *d_Wn, int64_t ldwn, const float *d_bn, const float *d_scale, const float *d_offset, int act,
                      float drop_p, uint64_t drop_seed, const float *d_dout, const float *d_dout_dropped, float *d_dX,
                      float *d_dWs, float *d_dWn, float *d_dbias, float *d_dscale, float *d_doffset, float *d_buf,
                      float *d_an_partial, float *d_tn_partial, void *d_pack, int dz_ready, const sl_sage_below *below,
                      float *d_dzs_amax, const uint32_t *d_dout_rows, uint32_t num_dout_rows, const float *d_x_amax,
                      const uint32_t *d_dout_map, void *stream);

/* A whole stack of chained GraphSAGE layers per call -- the conv loop of DeepGNN.forward (shaDow/models.py:193-197) over
 * GraphSAGE layers (layers.py:471-483) under residue 'none' + centre pooling (layers.py:159-163), forward and backward.  The
 * entries run sl_sage_fwd / sl_sage_bwd_chain layer by layer with exactly the arguments a caller of those would pass (layer l
 * reads layer l - 1's `out` / `out_amax`; backward top-down, each call's `below` = the next lower layer, its d_buf / d_dzs_amax
 * the halves of d_buf / d_amax in turn): the same kernels in the same order, identical results -- what they remove is the host
 * work between the layers, which bounds a step at the reference's own batch sizes (16-256 roots).
 * One descriptor per layer; the caller owns every buffer:
 *   parameters        Ws, Wn [Fout, Fin] (row pitch ldws / ldwn), bs, bn [Fout] (may be NULL), scale, offset [2, Fout]
 *   drop_p, drop_seed the layer's fused OUTPUT dropout (= the next layer's input dropout); 0 for the top layer
 *   forward products  AX [n, Fin] (pitch ldax), Zs, Zn, out [n, Fout] dense, out_amax [n], row_stats [n, 4] or NULL
 *                     (written by sl_sage_stack_fwd, read by sl_sage_stack_bwd)
 *   gradients         dWs, dWn [Fout, Fin], dbias [2, Fout] (NULL: no bias), dscale, doffset [2, Fout]
 * Fin of layer l = Fout of layer l - 1; every chained layer needs Fout % 32 == 0 (see sl_sage_bwd_chain).
 * sl_sage_stack_fwd: d_X0 [n, Fin_0] (pitch ldx0), d_x0_amax / x0_pad_zero as sl_sage_fwd's d_x_amax / x_pad_zero for layer 0;
 * d_pack: sl_sage_stack_pack_bytes(n, L, layers) bytes (shared by the layers: stream order).
 * sl_sage_stack_bwd: d_dout = the gradient of the top layer's `out`, [n, Fout] or -- d_dout_rows != NULL -- compact
 * [num_dout_rows, Fout] for the selected rows only (the read-out's feat[roots]); d_dX0 [n, Fin_0] dense or NULL (input gradient
 * not wanted); scratch: d_buf 2 * n * 3 * max Fout floats, d_amax 2 n floats, d_an_partial as for sl_act_norm_bwd (nb = 2),
 * d_chain_partial max_l sl_sage_chain_partial_floats(n, Fout_l) floats (L > 1), d_tn_partial the largest any layer's
 * sl_sage_bwd_chain needs, d_pack as above.  adj must carry the transposed adjacency when L > 1 or d_dX0 != NULL.          */
typedef struct {
  const float *Ws, *bs, *Wn, *bn, *scale, *offset;
  int64_t ldws, ldwn;
  uint32_t Fin, Fout;
  int act;
  float drop_p;
  uint64_t drop_seed;
  float *AX;
  int64_t ldax;
  float *Zs, *Zn, *out, *out_amax, *row_stats;
  float *dWs, *dWn, *dbias, *dscale, *doffset;
} sl_sage_stack_layer;
size_t sl_sage_stack_pack_bytes(uint32_t n, uint32_t L, const sl_sage_stack_layer *layers);
int sl_sage_stack_fwd(const sl_norm_adj *adj, const float *d_X0, int64_t ldx0, const float *d_x0_amax, int x0_pad_zero, uint32_t L,
                      const sl_sage_stack_layer *layers, void *d_pack, void *stream);
int sl_sage_stack_bwd(const sl_norm_adj *adj, const float *d_X0, int64_t ldx0, const float *d_x0_amax, uint32_t L,
                      const sl_sage_stack_layer *layers, const float *d_dout, const uint32_t *d_dout_rows, uint32_t num_dout_rows,
                      float *d_dX0, float *d_buf, float *d_amax, float *d_an_partial, float *d_chain_partial, float *d_tn_partial,
                      void *d_pack, void *stream);
/* The lower part of a stack whose top layers ran row-sparse (round 5): the chained dense passes of sl_sage_stack_bwd over the
 * layers 0 .. L - 1 of `layers`, where layer L - 1's [dZs | . | dZn] (d_top_buf, [n, 3 Fout]), its row maxima (d_top_amax) and its
 * dscale / doffset / dbias were already produced by the caller (sl_gemm_an_bwd_corr of the layer above).                      */
int sl_sage_stack_bwd_ready(const sl_norm_adj *adj, const float *d_X0, int64_t ldx0, const float *d_x0_amax, uint32_t L,
                            const sl_sage_stack_layer *layers, float *d_top_buf, const float *d_top_amax, float *d_dX0, float *d_buf,
                            float *d_amax, float *d_chain_partial, float *d_tn_partial, void *d_pack, void *stream);

/* One GCN layer pass per call (shaDow/layers.py:417-444 and its autograd):  out = norm(act((A X) W^T + b))  [+ the next
 * layer's input dropout / dual output as above].  forward: SpMM -> weight pack -> split-bf16 GEMM -> fused bias / act /
 * norm; the caller provides AX [n, Fin] (ldax), Z, out [n, Fout] and d_pack (sl_gcn_pack_bytes bytes).  backward:
 * act_norm backward -> dAX = dZ W -> dX = A^T dAX (d_dX NULL: not wanted; lddx its row pitch) -> dW = dZ^T AX; d_buf is
 * n * (Fout + Fin) floats of scratch, d_an_partial as for sl_act_norm_bwd (nb = 1), d_tn_partial as for sl_gemm_tn_f32.
 * Fout, Fin % 4 == 0, <= 256.  The same kernels in the same order as the separate entries: identical results.        */
size_t sl_gcn_pack_bytes(uint32_t n, uint32_t Fin, uint32_t Fout);
int sl_gcn_fwd(const sl_norm_adj *adj, const float *d_X, int64_t ldx, uint32_t Fin, uint32_t Fout, const float *d_W, int64_t ldw,
               const float *d_b, const float *d_scale, const float *d_offset, int act, float drop_p, uint64_t drop_seed,
               float *d_AX, int64_t ldax, float *d_Z, float *d_out, float *d_out_dropped, void *d_pack, void *stream);
int sl_gcn_bwd(const sl_norm_adj *adj, const float *d_AX, int64_t ldax, const float *d_Z, uint32_t Fin, uint32_t Fout,
               const float *d_W, int64_t ldw, const float *d_b, const float *d_scale, const float *d_offset, int act,
               float drop_p, uint64_t drop_seed, const float *d_dout, const float *d_dout_dropped, float *d_dX, int64_t lddx,
               float *d_dW, float *d_dbias, float *d_dscale, float *d_doffset, float *d_buf, float *d_an_partial,
               float *d_tn_partial, void *d_pack, void *stream);

/* A whole stack of GCN layers per call (as sl_sage_stack_* for GraphSAGE): sl_gcn_fwd / sl_gcn_bwd layer by layer with the arguments a
 * per-layer caller passes -- identical results.  Per layer: W [Fout, Fin] (pitch ldw), b [Fout] or NULL, scale, offset [Fout], the
 * fused output dropout (drop_p, drop_seed), the forward products AX [n, Fin] (pitch ldax), Z, out [n, Fout] dense, and the gradients dW
 * [Fout, Fin], dbias (NULL: no bias), dscale, doffset [Fout].  sl_gcn_stack_bwd: d_dout = the gradient of the top layer's `out`, dense
 * [n, Fout] or -- d_dout_rows != NULL -- [num_dout_rows, Fout] for the selected (distinct) rows, scattered into a cleared buffer;
 * d_dX0 [n, Fin_0] or NULL; scratch: d_grad 2 n max(F) floats (the layers' output gradients in turn), d_buf n * max(Fout + Fin)
 * floats, d_an_partial / d_tn_partial / d_pack as for sl_gcn_bwd (the largest any layer needs; sl_gcn_stack_pack_bytes).          */
typedef struct {
  const float *W, *b, *scale, *offset;
  int64_t ldw;
  uint32_t Fin, Fout;
  int act;
  float drop_p;
  uint64_t drop_seed;
  float *AX;
  int64_t ldax;
  float *Z, *out;
  float *dW, *dbias, *dscale, *doffset;
} sl_gcn_stack_layer;
size_t sl_gcn_stack_pack_bytes(uint32_t n, uint32_t L, const sl_gcn_stack_layer *layers);
int sl_gcn_stack_fwd(const sl_norm_adj *adj, const float *d_X0, int64_t ldx0, uint32_t L, const sl_gcn_stack_layer *layers, void *d_pack,
                     void *stream);
int sl_gcn_stack_bwd(const sl_norm_adj *adj, uint32_t L, const sl_gcn_stack_layer *layers, const float *d_dout,
                     const uint32_t *d_dout_rows, uint32_t num_dout_rows, float *d_dX0, float *d_grad, float *d_buf, float *d_an_partial,
                     float *d_tn_partial, void *d_pack, void *stream);

/* Head of a node-classification step (csrc/head.hip): emb [r, F] = the roots' rows of the last layer (layers.py:159-163) ->
 * xn = emb / max(|emb|_2, 1e-12) (models.py:200, F.normalize) -> z = xn W^T + b, preds = _f_norm_feat(z) over the C classes
 * (the one-layer classifier MLP(dim_hid -> num_classes, act 'I'), models.py:139-146, layers.py:329-338, 376-400) -> prob =
 * softmax(preds), loss = mean_i CE(preds_i, label_i) (models.py:163-166) -- one kernel + a one-workgroup mean; backward
 * (sl_head_bwd) two kernels + a slice reduction: d_gloss [1] (device) = d L / d loss, outputs d_demb [r, F], d_dW [C, F] dense,
 * d_db, d_dscale, d_doffset [C].  F % 4 == 0 <= 256, C <= 256, 16-byte aligned rows; d_label int64 class indices; d_xn [r, F],
 * d_z / d_preds / d_prob [r, C] dense, d_nrm, d_rowloss [r], d_loss [1]; d_work 3 r C floats, d_partial
 * sl_head_partial_floats(r, F, C) floats.  The sums over the roots run in a fixed order (run-to-run identical results).       */
size_t sl_head_partial_floats(uint32_t r, uint32_t F, uint32_t C);
int sl_head_fwd(const float *d_emb, int64_t lde, const float *d_W, int64_t ldw, const float *d_b, const float *d_scale,
                const float *d_offset, const int64_t *d_label, uint32_t r, uint32_t F, uint32_t C, float *d_xn, float *d_z,
                float *d_preds, float *d_prob, float *d_nrm, float *d_rowloss, float *d_loss, void *stream);
int sl_head_bwd(const float *d_gloss, const float *d_xn, const float *d_z, const float *d_prob, const float *d_nrm,
                const int64_t *d_label, const float *d_W, int64_t ldw, const float *d_scale, uint32_t r, uint32_t F, uint32_t C,
                float *d_demb, float *d_dW, float *d_db, float *d_dscale, float *d_doffset, float *d_work, float *d_partial,
                void *stream);

/* Per-kernel timing inside the layer entries above (csrc/prof.hip): while sl_prof_enable(1) is in effect every kernel a
 * sl_sage_* / sl_gcn_* entry launches is bracketed by a HIP-event pair on its stream, under the kernel class names and
 * algorithmic byte / flop counts bench.py reports -- the measured call path is the timed call
 * path.  sl_prof_enable(1) clears earlier records; (0) stops recording; a negative argument only queries; returns the
 * previous state.  sl_prof_dump waits for the events and writes one line per kernel class,
 * "name\tlaunches\ttimed launches\ttotal ms\tbytes\tflops\n" (sums over the timed launches), into buf (at most cap
 * bytes incl. the terminating 0); returns the size the text needs.                                                */
int sl_prof_enable(int on);
size_t sl_prof_dump(char *buf, size_t cap);

/* End of a training step on flat fp32 buffers (shaDow/models.py:225-226: torch.nn.utils.clip_grad_norm_(parameters, 5)
 * + torch.optim.Adam.step(), default betas / eps, no weight decay): the gradient is scaled IN PLACE by
 * min(1, max_norm / (||g|| + 1e-6)) (max_norm <= 0: no clipping), the moments and the parameters are updated with
 * torch's arithmetic (the hyper-parameters are doubles, as in Python: 1 - beta and lr / (1 - beta^t) are formed in double
 * precision before they are rounded); `step` counts from 1.  Two launches (norm partials, update), deterministic.  d_scratch:
 * sl_clip_adam_scratch_floats() floats; its last float receives ||g|| before clipping.  Buffers 16-byte aligned.     */
uint32_t sl_clip_adam_scratch_floats(void);
/* Zeros into the F-wide column slices d_a and d_b (either may be NULL) of a buffer with row pitch ld: the rows around the
 * few a row-sparse backward pass fills itself (F % 4 == 0, ld % 4 == 0, 16-byte aligned slices). */
int sl_zero_slices(float *d_a, float *d_b, int64_t ld, uint32_t n, uint32_t F, void *stream);

/* Row sets and input gradient of the exact row-sparse backward pass of a stack's top layer under a read-out that takes ONE
 * row per subgraph (the roots, shaDow/layers.py:159-163): the layer's dZ is zero outside the roots R and its input gradient
 * zero outside T = R u N(R), where a row j gets at most one neighbour term -- from its own subgraph's root through the
 * edge (root, j).  sl_top_plan: d_targets [P] = the root row of every subgraph (num_roots = 1, ascending); writes
 * d_off [P + 2] (exclusive offsets of the subgraphs' shares of T, d_off[P] = |T|, d_off[P + 1] != 0: a root's neighbour list
 * is not strictly ascending -- repeated edges -- and the plan must not be used), and, when |T| <= cap, d_T / d_slot / d_epos
 * [|T|] (row id, subgraph, position of the edge (root, row) in the batch CSR or -1 for a root without self edge) and
 * d_self_idx [P] (position of every root in T).  sl_top_dx: d_out[j, :] = A[root, T[j]] G[slot[j], :] + [T[j] is the root]
 * S[slot[j], :] with A = diag(row_scale) (A o edge_w) diag(col_scale) (any of the three may be NULL), G = dZn[R] Wn,
 * S = dZs[R] Ws ([P, F] each, pitch ldg); F % 4 == 0, 16-byte aligned rows.                                            */
int sl_top_plan(const uint32_t *d_indptr, const uint32_t *d_indices, const uint32_t *d_targets, uint32_t num_subg, uint32_t cap,
                uint32_t *d_off, uint32_t *d_T, uint32_t *d_slot, int32_t *d_epos, uint32_t *d_self_idx, void *stream);
/* The row map of T and the batch's TRANSPOSED adjacency restricted to the columns T, behind sl_top_plan on the same stream (no
 * host round trip in between): d_rowmap[i] = position of row i in T, 0xFFFFFFFF outside; (d_f_indptr [n + 1], d_f_indices,
 * d_f_perm [e]) = per row of A^T the entries whose source row lies in T, in their original order, column ids = positions in T,
 * d_f_perm = the edge's place in the batch CSR.  d_work: n / 1024 + 4 words.  The number of kept entries lands in
 * d_off[num_subg + 2] (d_off: the plan's [num_subg + 3] counter array).  The aggregation A^T dZ of a gradient that lives on T
 * runs over this structure (sl_spmm_csr_amax_f32 on the compact [t, F] rows).                                           */
int sl_top_plan_filter(const uint32_t *d_T, uint32_t *d_off, uint32_t num_subg, uint32_t cap, const uint32_t *d_t_indptr,
                       const uint32_t *d_t_indices, const uint32_t *d_t_perm, uint32_t n, uint32_t *d_rowmap, uint32_t *d_f_indptr,
                       uint32_t *d_f_indices, uint32_t *d_f_perm, uint32_t *d_work, void *stream);
int sl_top_dx(const float *d_G, const float *d_S, int64_t ldg, const uint32_t *d_T, const uint32_t *d_slot, const int32_t *d_epos,
              const uint32_t *d_self_idx, const uint32_t *d_targets, const float *d_edge_w, const float *d_row_scale,
              const float *d_col_scale, uint32_t t, uint32_t F, float *d_out, int64_t ldo, void *stream);

int sl_clip_adam(float *d_param, float *d_grad, float *d_exp_avg, float *d_exp_avg_sq, uint64_t n, double lr, double beta1,
                 double beta2, double eps, uint32_t step, float max_norm, float *d_scratch, void *stream);

/* Fused multi-head GAT attention aggregate (GAT._aggregate_attention for all
 * heads, shaDow/layers.py:560-582,612-619):
 *   hn = act(z_neigh);  u_s = att[0,h]·act(z_self)_h;  u_n = att[1,h]·hn_h
 *   e_ij = lrelu0.2(u_s[i]) + lrelu0.2(u_n[j]);  row softmax with max
 *   subtraction, numerator * edge_w (drop-edge mask), denominator >= 1e-10
 *   nagg_i = sum_j p_ij hn_j / den_i
 * F = heads*D <= 256, D = 4*2^k.  Outputs hn[n,F], u_s/u_n/mx/den[n,heads]
 * are kept for the backward pass.  d_hn may be NULL (forward and backward alike): hn is then not materialised, the
 * kernels apply the activation to the z_neigh rows they gather.                                                  */
int sl_gat_fwd(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w,
               const float *d_z_self, const float *d_z_neigh, const float *d_att, int act, uint32_t n,
               uint32_t F, uint32_t heads, float *d_hn, float *d_u_s, float *d_u_n, float *d_mx,
               float *d_den, float *d_nagg, void *stream);
/* The row pass of sl_gat_fwd alone: d_hn, d_u_s, d_u_n are given (sl_gemm_nt2_gat_f32 left them).  sl_gat_bwd then takes
 * d_z_neigh = NULL: the activation's derivative follows from hn (relu / elu / leaky relu: z > 0 <=> hn > 0; tanh: 1 - hn^2). */
int sl_gat_fwd_rows(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w, const float *d_hn, const float *d_u_s,
                    const float *d_u_n, uint32_t n, uint32_t F, uint32_t heads, float *d_mx, float *d_den, float *d_nagg, void *stream);
/* sl_gat_fwd_rows and the layer's activation + per-head feature normalisation + branch average + output dropout in the same
 * pass (shaDow/layers.py:612-625, 329-338): the aggregate of a row leaves the kernel normalised,
 *   d_out = out_scale * (norm_0(N) + norm_1(act(z_self)))   per head slice (d_scale / d_offset [2, F]: row 0 the aggregate's)
 * followed by the fused output dropout (drop_p, drop_seed: the mask rule of sl_act_norm_fwd) and, optionally, the row maxima
 * d_out_amax [n] -- what sl_act_norm_fwd(nb = 2, seg = F / heads) makes of (N, z_self), to rounding (the compiler fuses
 * multiply-add pairs differently in the two kernels: 1 - 2 ulp per stage; same dropout mask).  d_nagg (may be NULL when
 * no backward pass follows) still receives N.                                                                               */
int sl_gat_fwd_tail(const uint32_t *d_indptr, const uint32_t *d_indices, const float *d_edge_w, const float *d_hn,
                    const float *d_u_s, const float *d_u_n, const float *d_z_self, int act, const float *d_scale,
                    const float *d_offset, uint32_t n, uint32_t F, uint32_t heads, float out_scale, float drop_p,
                    uint64_t drop_seed, float *d_mx, float *d_den, float *d_nagg, float *d_out, float *d_out_amax, void *stream);
/* Backward: dz_self (attention part only), dz_neigh, datt[2,heads,D].
 * Round 6: ONE edge walk.  The column walk (transposed CSR) gathers dN_i for every edge (i -> j) anyway and holds hn_j in
 * registers: it forms alpha_ij again from row i's (u_s, max, denominator) -- the forward pass's expression -- and
 * de_ij = alpha_ij (dN_i . hn_j - t_i) itself; the row walk that gathered hn_j per edge to leave alpha / de [e, heads] behind is
 * gone.  d_t [n, heads]: t_i = dN_i . N_i per head, left by the act + norm backward that produced d_dnagg
 * (sl_act_norm_bwd_rows_t); NULL: computed here into d_work.
 * d_work: float[n*heads + 4096*F] (t when not given, and the per-block partial sums of datt, which are added in block order:
 * no float atomics, bit-reproducible).                */
/* The attention's share of dz_self and datt[0] are EXACTLY zero: a row's weights exp(e_ij - max_j e_ij) / max(sum, 1e-10) do not
 * change when u_s[i] moves every e_ij of the row together (on a row whose sum sits on the clamp the reference's autograd gets
 * there through the arg-max of torch_scatter's max; the column walk hands -t_i to that edge).  d_z_self is not read (may be NULL).
 * accumulate_dz_self != 0: d_dz_self already holds the share of z_self's gradient that came through the layer's own act_norm
 * branch and is left alone; 0: it is cleared.  d_row_amax (may be NULL): receives max_k over the final dz_self AND dz_neigh rows
 * (the operand scale of the K-concatenated input-gradient product, sl_gemm_nt_cat_f32); with accumulate_dz_self != 0 it must
 * come in HOLDING max_k |dz_self[i, k]| of the incoming rows (an upper bound will do).                     */
int sl_gat_bwd(const uint32_t *d_indptr, const uint32_t *d_indices, const uint32_t *d_t_indptr,
               const uint32_t *d_t_indices, const uint32_t *d_t_perm, const float *d_edge_w,
               const float *d_z_self, const float *d_z_neigh, const float *d_att, int act, uint32_t n,
               uint32_t e, uint32_t F, uint32_t heads, const float *d_hn, const float *d_u_s,
               const float *d_u_n, const float *d_mx, const float *d_den, const float *d_nagg,
               const float *d_dnagg, const float *d_t, float *d_work, float *d_dz_self, float *d_dz_neigh, float *d_datt,
               int accumulate_dz_self, float *d_row_amax, void *stream);
/* (ABI 25) sl_gat_bwd for an incoming gradient that lives on a FEW rows -- the layer below a row-sparse backward pass
 * (tail.build_backward_levels): d_dnagg_rows [t, F] and d_t_rows [t, heads] are compact, row i's at d_dn_map[i] (uint32 [n],
 * 0xFFFFFFFF: dN_i = 0).  The column walk passes over the edges of such rows -- their terms are alpha_ij * 0 and 0, exactly what
 * sl_gat_bwd adds for a zero row: bit-identical to sl_gat_bwd on the expanded [n, F] / [n, heads] tensors -- and the expansion is
 * never written.  d_t_rows is required (sl_act_norm_bwd_rows_t with dz_compact leaves it).                                   */
int sl_gat_bwd_map(const uint32_t *d_indptr, const uint32_t *d_indices, const uint32_t *d_t_indptr,
                   const uint32_t *d_t_indices, const uint32_t *d_t_perm, const float *d_edge_w,
                   const float *d_z_self, const float *d_z_neigh, const float *d_att, int act, uint32_t n,
                   uint32_t e, uint32_t F, uint32_t heads, const float *d_hn, const float *d_u_s,
                   const float *d_u_n, const float *d_mx, const float *d_den, const float *d_nagg,
                   const float *d_dnagg_rows, const float *d_t_rows, const uint32_t *d_dn_map, float *d_work, float *d_dz_self,
                   float *d_dz_neigh, float *d_datt, int accumulate_dz_self, float *d_row_amax, void *stream);

/* Development aid: per-subgraph result words of the last sg_sample call,
 * 16 uint32 per subgraph: {nodes, edges, flags, stream slots, frontier nodes,
 * frontier reads, -, -, 8 x phase cycle stamps (only when the library was
 * built with -DSHADOW_SG_TIMING)}.  Synchronises the device.                  */
int sg_debug_subgraph_stats(sg_sampler *s, uint32_t *h_out, uint32_t max_subgraphs);
/* Debug: the 16 plan words of the last sg_sample call -- [0] scan work items, [1] chunks per item, [8..12] cycles/16
 * the scan workgroups' leaders spent in (item setup, chunk start rows, scan, candidate resolution, sort + write),
 * [13] items, [14] rounds.  Synchronises the device.                                                          */
int sg_debug_scan_phases(sg_sampler *s, uint32_t *h_out16);
/* Calibration: stream the full-graph rows of `n` nodes (device array d_nodes) as the scan does (aligned 16-byte quads,
 * one wavefront per row, `depth` in {1, 2, 4} 1-KiB loads in flight, `blocks` workgroups of 256 threads) and fold each
 * row into d_out[i]: the memory system's ceiling for the sampler's access pattern (scripts/probe_row_stream.py).   */
int sg_debug_stream_rows(sg_sampler *s, const uint32_t *d_nodes, uint32_t n, uint32_t *d_out, int depth, int blocks,
                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SHADOW_HIP_H */
