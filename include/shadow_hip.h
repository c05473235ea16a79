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

/* Development aid: per-subgraph result words of the last sg_sample call,
 * 16 uint32 per subgraph: {nodes, edges, flags, stream slots, frontier nodes,
 * frontier reads, -, -, 8 x phase cycle stamps (only when the library was
 * built with -DSHADOW_SG_TIMING)}.  Synchronises the device.                  */
int sg_debug_subgraph_stats(sg_sampler *s, uint32_t *h_out, uint32_t max_subgraphs);

#ifdef __cplusplus
}
#endif
#endif /* SHADOW_HIP_H */
