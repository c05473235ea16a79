"""ctypes binding of libshadow_hip.so (the C ABI declared in include/shadow_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).
There is NO CPU fallback: if the shared object is missing, or a call returns a
non-zero status, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libshadow_hip.so")

SG_OK, SG_ERR_INVALID, SG_ERR_HIP, SG_ERR_CAPACITY, SG_ERR_IO, SG_ERR_STATE = 0, 1, 2, 3, 4, 5
SG_METHOD = {"khop": 0, "ppr": 1, "nodeIID": 2}
SG_AUG = {"hops": 1, "pprs": 2, "drnls": 4}


class ShadowHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libshadow_hip status {code}: {msg}")
        self.code = code


class CapacityError(ShadowHipError):
    """SG_ERR_CAPACITY -- grow the named capacity and call again."""


class SgConfig(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("num_roots", C.c_int32), ("depth", C.c_int32),
        ("budget", C.c_int32), ("k", C.c_int32), ("threshold", C.c_float),
        ("add_self_edge", C.c_int32), ("include_target_conn", C.c_int32),
        ("compat_overread", C.c_int32), ("aug_flags", C.c_int32),
    ]


class SgBatchOut(C.Structure):
    _fields_ = [
        ("d_node", C.c_void_p), ("d_indptr", C.c_void_p), ("d_indices", C.c_void_p),
        ("d_edge_id", C.c_void_p), ("d_target", C.c_void_p), ("d_subg_nodes", C.c_void_p),
        ("d_subg_edges", C.c_void_p), ("d_hop", C.c_void_p), ("d_ppr", C.c_void_p),
        ("d_drnl", C.c_void_p), ("cap_nodes", C.c_uint64), ("cap_edges", C.c_uint64),
    ]


class SgBatchCounts(C.Structure):
    _fields_ = [
        ("n_tot", C.c_uint64), ("e_tot", C.c_uint64), ("num_subgraphs", C.c_uint32),
        ("max_subg_nodes", C.c_uint32), ("max_subg_edges", C.c_uint32), ("overflow", C.c_uint32),
        ("slots_scanned", C.c_uint64), ("frontier_reads", C.c_uint64), ("frontier_nodes", C.c_uint64),
        ("sample_kernel_ms", C.c_float), ("relocate_kernel_ms", C.c_float),
    ]


class SlSageBelow(C.Structure):
    """sl_sage_below: the GraphSAGE layer whose act_norm backward rides in the layer above's input-gradient product."""
    _fields_ = [
        ("Zs", C.c_void_p), ("Zn", C.c_void_p), ("bs", C.c_void_p), ("bn", C.c_void_p), ("scale", C.c_void_p),
        ("offset", C.c_void_p), ("act", C.c_int), ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("F", C.c_uint32),
        ("buf", C.c_void_p), ("dscale", C.c_void_p), ("doffset", C.c_void_p), ("dbias", C.c_void_p), ("partial", C.c_void_p),
        ("amax", C.c_void_p), ("stats", C.c_void_p), ("dout_plain", C.c_void_p), ("plain_row", C.c_void_p), ("plain_ld", C.c_int64),
    ]


class SlRowsJob(C.Structure):
    """sl_rows_job: one row copy of sl_rows_multi (mode 0 gather, 1 clear, 2 scatter)."""
    _fields_ = [("src", C.c_void_p), ("lds", C.c_int64), ("dst", C.c_void_p), ("ldd", C.c_int64), ("width", C.c_uint32), ("mode", C.c_int)]


class SlSageStackLayer(C.Structure):
    """sl_sage_stack_layer: one GraphSAGE layer of a stack run by sl_sage_stack_fwd / sl_sage_stack_bwd."""
    _fields_ = [
        ("Ws", C.c_void_p), ("bs", C.c_void_p), ("Wn", C.c_void_p), ("bn", C.c_void_p), ("scale", C.c_void_p), ("offset", C.c_void_p),
        ("ldws", C.c_int64), ("ldwn", C.c_int64), ("Fin", C.c_uint32), ("Fout", C.c_uint32), ("act", C.c_int), ("drop_p", C.c_float),
        ("drop_seed", C.c_uint64), ("AX", C.c_void_p), ("ldax", C.c_int64), ("Zs", C.c_void_p), ("Zn", C.c_void_p), ("out", C.c_void_p),
        ("out_amax", C.c_void_p), ("row_stats", C.c_void_p), ("dWs", C.c_void_p), ("dWn", C.c_void_p), ("dbias", C.c_void_p),
        ("dscale", C.c_void_p), ("doffset", C.c_void_p),
    ]


class SlGcnStackLayer(C.Structure):
    """sl_gcn_stack_layer: one GCN layer of a stack run by sl_gcn_stack_fwd / sl_gcn_stack_bwd."""
    _fields_ = [
        ("W", C.c_void_p), ("b", C.c_void_p), ("scale", C.c_void_p), ("offset", C.c_void_p), ("ldw", C.c_int64), ("Fin", C.c_uint32),
        ("Fout", C.c_uint32), ("act", C.c_int), ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("AX", C.c_void_p), ("ldax", C.c_int64),
        ("Z", C.c_void_p), ("out", C.c_void_p), ("dW", C.c_void_p), ("dbias", C.c_void_p), ("dscale", C.c_void_p), ("doffset", C.c_void_p),
    ]


class SlNormAdj(C.Structure):
    _fields_ = [
        ("indptr", C.c_void_p), ("indices", C.c_void_p), ("edge_w", C.c_void_p), ("row_scale", C.c_void_p),
        ("col_scale", C.c_void_p), ("t_indptr", C.c_void_p), ("t_indices", C.c_void_p), ("t_perm", C.c_void_p),
        ("subg_node_off", C.c_void_p), ("subg_edge_off", C.c_void_p), ("num_subg", C.c_uint32),
        ("max_subg_nodes", C.c_uint32), ("n", C.c_uint32), ("e", C.c_uint32), ("row_entries_bound", C.c_uint32),
    ]


# name -> (restype, argtypes); every symbol declared in include/shadow_hip.h
_P = C.c_void_p
SIGNATURES = {
    "sg_last_error": (C.c_char_p, []),
    "sg_abi_version": (C.c_int, []),
    "sg_create": (C.c_int, [_P, _P, C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.POINTER(_P)]),
    "sg_create_from_bin": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int64, C.POINTER(_P)]),
    "sg_create_from_bin_ex": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(_P)]),
    "sg_destroy": (None, [_P]),
    "sg_num_nodes": (C.c_uint32, [_P]),
    "sg_num_edges": (C.c_uint64, [_P]),
    "sg_num_nodes_target": (C.c_uint64, [_P]),
    "sg_get_idx_root": (C.c_uint64, [_P]),
    "sg_device_indptr": (_P, [_P]),
    "sg_device_indices": (_P, [_P]),
    "sg_shuffle_targets": (C.c_int, [_P, _P, C.c_uint64]),
    "sg_next_roots": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "sg_set_caps": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "sg_get_caps": (C.c_int, [_P, C.POINTER(SgConfig), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sg_set_ppr": (C.c_int, [_P, _P, C.c_uint32, _P, _P, _P, C.c_uint32]),
    "sg_load_ppr_bin": (C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_float]),
    "sg_save_ppr_bin": (C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_float]),
    "sg_drop_full_graph_info": (C.c_int, [_P]),
    "sg_ppr_push": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _P,
                               C.c_uint64, _P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint32, _P]),
    "sg_sample": (C.c_int, [_P, C.POINTER(SgConfig), C.c_uint64, C.c_uint32, C.c_uint64, _P,
                             C.POINTER(SgBatchOut), _P]),
    "sg_sample_finish": (C.c_int, [_P, C.POINTER(SgBatchCounts)]),
    "sg_sample_multi": (C.c_int, [_P, C.POINTER(SgConfig), C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64, _P,
                                  C.POINTER(SgBatchOut), _P]),
    "sg_sample_finish_multi": (C.c_int, [_P, C.c_uint32, C.POINTER(SgBatchCounts)]),
    "sg_set_profiling": (C.c_int, [_P, C.c_int]),
    "sg_debug_subgraph_stats": (C.c_int, [_P, _P, C.c_uint32]),
    "sg_debug_scan_phases": (C.c_int, [_P, _P]),
    "sg_debug_stream_rows": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_int, C.c_int, _P]),
    "sl_gather_rows_f32": (C.c_int, [_P, C.c_int64, _P, C.c_uint32, C.c_uint32, _P, C.c_int64, _P]),
    "sl_rows_multi": (C.c_int, [C.POINTER(SlRowsJob), C.c_int, _P, C.c_uint32, _P]),
    "sl_gather_rows_drop_f32": (C.c_int, [_P, C.c_int64, _P, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, _P, C.c_int64,
                                           C.c_uint32, _P, _P]),
    "sl_csr_edge_rows": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P]),
    "sl_csr_transpose": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P]),
    "sl_degree_scales": (C.c_int, [_P, _P, C.c_uint32, C.c_int, _P, _P]),
    "sl_spmm_csr_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int64, C.c_uint32,
                                   C.c_uint32, _P]),
    "sl_spmm_csr_amax_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int64, C.c_uint32,
                                        C.c_uint32, _P, _P]),
    "sl_merge_subgraphs": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "sl_spmm_blockdiag_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int64, C.c_uint32,
                                         C.c_uint32, _P, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "sl_spmm_blockdiag_gather_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_float, C.c_uint64, _P, C.c_int64,
                                                _P, C.c_int64, C.c_uint32, C.c_uint32, _P, _P, C.c_uint32, C.c_uint32, _P]),
    "sg_cache_create": (C.c_int, [C.c_uint32, C.c_int, C.POINTER(_P)]),
    "sg_cache_destroy": (None, [_P]),
    "sg_cache_clear": (C.c_int, [_P]),
    "sg_cache_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "sg_cache_record": (C.c_int, [_P, _P, C.c_uint32, C.c_uint64, C.c_uint64, _P]),
    "sg_cache_collate": (C.c_int, [_P, _P, C.c_uint32, _P, _P]),
    "sg_cache_collate_finish": (C.c_int, [_P, _P]),
    "sl_gemm_pack_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "sl_gemm_pack_b": (C.c_int, [_P, C.c_int64, C.c_uint32, C.c_uint32, _P, _P]),
    "sl_gemm_nt_f32": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    "sl_gemm_pack_b2": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_uint32, _P, C.c_int64, C.c_int64, C.c_uint32, C.c_uint32, _P, _P]),
    "sl_sage_pack_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "sl_sage_fwd": (C.c_int, [C.POINTER(SlNormAdj), _P, C.c_int64, C.c_uint32, C.c_uint32, _P, C.c_int64, _P, _P, C.c_int64, _P, _P, _P,
                               C.c_int, C.c_float, C.c_uint64, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "sl_sage_bwd": (C.c_int, [C.POINTER(SlNormAdj), _P, C.c_int64, _P, C.c_int64, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_int64, _P,
                               _P, C.c_int64, _P, _P, _P, C.c_int, C.c_float, C.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                               _P, _P]),
    "sl_sage_chain_partial_floats": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "sl_sage_bwd_chain": (C.c_int, [C.POINTER(SlNormAdj), _P, C.c_int64, _P, C.c_int64, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_int64, _P,
                                     _P, C.c_int64, _P, _P, _P, C.c_int, C.c_float, C.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                     _P, C.c_int, C.POINTER(SlSageBelow), _P, _P, C.c_uint32, _P, _P, _P]),
    "sl_sage_stack_pack_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32, C.POINTER(SlSageStackLayer)]),
    "sl_sage_stack_fwd": (C.c_int, [C.POINTER(SlNormAdj), _P, C.c_int64, _P, C.c_int, C.c_uint32, C.POINTER(SlSageStackLayer), _P, _P]),
    "sl_sage_stack_bwd": (C.c_int, [C.POINTER(SlNormAdj), _P, C.c_int64, _P, C.c_uint32, C.POINTER(SlSageStackLayer), _P, _P, C.c_uint32,
                                     _P, _P, _P, _P, _P, _P, _P, _P]),
    "sl_sage_stack_bwd_ready": (C.c_int, [C.POINTER(SlNormAdj), _P, C.c_int64, _P, C.c_uint32, C.POINTER(SlSageStackLayer), _P, _P,
                                           _P, _P, _P, _P, _P, _P, _P]),
    "sl_gcn_stack_pack_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32, C.POINTER(SlGcnStackLayer)]),
    "sl_gcn_stack_fwd": (C.c_int, [C.POINTER(SlNormAdj), _P, C.c_int64, C.c_uint32, C.POINTER(SlGcnStackLayer), _P, _P]),
    "sl_gcn_stack_bwd": (C.c_int, [C.POINTER(SlNormAdj), C.c_uint32, C.POINTER(SlGcnStackLayer), _P, _P, C.c_uint32, _P, _P, _P, _P, _P, _P,
                                    _P]),
    "sl_head_partial_floats": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "sl_head_fwd": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P, _P,
                               _P]),
    "sl_head_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P, _P,
                               _P]),
    "sl_set_fused_epilogue": (C.c_int, [C.c_int]),
    "sl_set_spmm_wide_pipe": (C.c_int, [C.c_int]),
    "sl_prof_enable": (C.c_int, [C.c_int]),
    "sl_prof_dump": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "sl_gemm_act_norm_supported": (C.c_int, [C.c_uint32, C.c_uint32]),
    "sl_gemm_act_norm_tiles": (C.c_uint32, [C.c_uint32]),
    "sl_gemm_act_norm_pack_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "sl_row_amax": (C.c_int, [_P, C.c_int64, C.c_uint32, C.c_uint32, _P, _P]),
    "sl_gemm_act_norm_pack": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(C.c_int64), C.c_uint32, C.c_uint32, _P, _P, C.c_uint32, _P]),
    "sl_gemm_act_norm_pack_b2": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_uint32, _P, C.c_int64, C.c_int64, C.c_uint32, C.c_uint32, _P, _P]),
    "sl_gemm_act_norm_fwd": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), _P, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), C.POINTER(C.c_int), _P, _P, C.c_float, _P,
                                        C.c_int64, C.c_float, C.c_uint64, _P, C.c_int64, _P, _P, _P]),
    "sl_gemm_nt2_f32": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), _P, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64), _P]),
    "sl_gemm_nt2_gat_supported": (C.c_int, [C.c_uint32, C.c_uint32]),
    "sl_gemm_nt2_gat_f32": (C.c_int, [_P, C.c_int64, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_P), _P, C.c_int64, _P, C.c_int64,
                                       _P, C.c_int, C.c_uint32, _P, _P, _P]),
    "sl_gemm_nt_cat_f32": (C.c_int, [_P, C.c_int64, C.c_uint32, _P, C.c_int64, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P,
                                      C.c_int64, _P]),
    "sl_gemm_an_bwd_partial_floats": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_int]),
    "sl_gemm_an_bwd": (C.c_int, [_P, C.c_int64, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64),
                                  C.POINTER(_P), C.POINTER(C.c_int), _P, _P, C.c_float, C.POINTER(_P), C.POINTER(C.c_int64), _P, _P, _P, _P,
                                  C.c_float, C.c_uint64, _P, _P, _P]),
    "sl_gemm_an_bwd_corr": (C.c_int, [_P, C.c_int64, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64),
                                  C.POINTER(_P), C.POINTER(C.c_int), _P, _P, C.c_float, C.POINTER(_P), C.POINTER(C.c_int64), _P, _P, _P, _P,
                                  C.c_float, C.c_uint64, _P, _P, _P, C.c_int64, _P, C.c_uint32, _P]),
    "sl_gemm_an_bwd_plain": (C.c_int, [_P, C.c_int64, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64),
                                  C.POINTER(_P), C.POINTER(C.c_int), _P, _P, C.c_float, C.POINTER(_P), C.POINTER(C.c_int64), _P, _P, _P, _P,
                                  C.c_float, C.c_uint64, _P, _P, _P, C.c_int64, _P, C.c_uint32, _P, C.c_int64, _P, _P]),
    "sl_gcn_pack_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "sl_gcn_fwd": (C.c_int, [C.POINTER(SlNormAdj), _P, C.c_int64, C.c_uint32, C.c_uint32, _P, C.c_int64, _P, _P, _P, C.c_int, C.c_float,
                              C.c_uint64, _P, C.c_int64, _P, _P, _P, _P, _P]),
    "sl_gcn_bwd": (C.c_int, [C.POINTER(SlNormAdj), _P, C.c_int64, _P, C.c_uint32, C.c_uint32, _P, C.c_int64, _P, _P, _P, C.c_int,
                              C.c_float, C.c_uint64, _P, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sl_clip_adam_scratch_floats": (C.c_uint32, []),
    "sl_clip_adam": (C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_float, _P, _P]),
    "sl_gemm_tn_slices": (C.c_uint32, [C.c_uint32]),
    "sl_gemm_tn_f32": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "sl_gemm_tn_f16": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "sl_gemm_tn_f16_pair": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int64, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P, _P]),
    "sl_segment_pool_fwd": (C.c_int, [_P, C.c_int64, _P, C.c_uint32, C.c_uint32, C.c_int, _P, C.c_int64, _P, _P]),
    "sl_segment_pool_bwd": (C.c_int, [_P, C.c_int64, _P, C.c_uint32, C.c_uint32, C.c_int, _P, _P, C.c_int64, _P]),
    "sl_pool_grad_rows": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "sl_pool_grad_table": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _P,
                                     C.c_int64, _P]),
    "sl_encode_codes": (C.c_int, [C.c_int, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "sl_onehot_linear_fwd": (C.c_int, [_P, C.c_int64, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_int64, _P]),
    "sl_onehot_linear_bwd": (C.c_int, [_P, C.c_int64, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P, C.c_uint32, _P]),
    "sl_act_norm_vector_layout": (C.c_int, [C.c_uint32, C.c_uint32]),
    "sl_act_norm_fwd": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), C.POINTER(C.c_int), _P, _P,
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, _P, C.c_int64, C.c_float, C.c_uint64, _P, C.c_int64, _P, _P]),
    "sl_gat_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                              _P, _P, _P, _P, _P, _P, _P]),
    "sl_gat_fwd_rows": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P, _P]),
    "sl_gat_fwd_tail": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                   C.c_uint64, _P, _P, _P, _P, _P, _P]),
    "sl_gat_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                              C.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "sl_gat_bwd_map": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "sl_top_plan": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    "sl_top_plan_filter": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    "sl_top_dx": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_int64, _P]),
    "sl_spmm_blockdiag_rows_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, _P, _P, C.c_int64, C.c_uint32, C.c_uint32, _P, _P,
                                              C.c_uint32, C.c_uint32, _P, C.c_uint32, _P]),
    "sl_zero_slices": (C.c_int, [_P, _P, C.c_int64, C.c_uint32, C.c_uint32, _P]),
    "sl_act_norm_bwd": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), C.POINTER(C.c_int), _P, _P,
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, _P, C.c_int64,
                                   C.POINTER(_P), C.POINTER(C.c_int64), _P, _P, _P, _P, C.c_float, C.c_uint64, _P, C.c_int64, _P, _P, _P]),
    "sl_act_norm_bwd_rows": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), C.POINTER(C.c_int), _P, _P,
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, _P, C.c_int64,
                                   C.POINTER(_P), C.POINTER(C.c_int64), _P, _P, _P, _P, C.c_float, C.c_uint64, _P, C.c_int64, _P, _P, C.c_int, _P]),
    "sl_act_norm_bwd_rows_t": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), C.POINTER(C.c_int), _P, _P,
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, _P, C.c_int64,
                                   C.POINTER(_P), C.POINTER(C.c_int64), _P, _P, _P, _P, C.c_float, C.c_uint64, _P, C.c_int64, _P, _P, C.c_int,
                                   C.c_int, _P, C.c_int, _P]),
    "sl_act_norm_bwd_map": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), C.POINTER(C.c_int), _P, _P,
                                      C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, _P, C.c_int64, _P,
                                      C.POINTER(_P), C.POINTER(C.c_int64), _P, _P, _P, _P, C.c_float, C.c_uint64, _P, C.c_int64, _P, _P]),
}

_lib = None


MAX_BATCHES_PER_CALL = 16      # SG_MAX_BATCHES_PER_CALL of include/shadow_hip.h
ABI_VERSION = 25      # sg_abi_version() of the library these signatures describe


def load():
    """Load libshadow_hip.so.  torch is imported first so that the HIP runtime
    torch ships (same SONAME libamdhip64.so.7) is the one the library binds to:
    one runtime per process, torch's streams and allocations are directly usable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root. "
            "shadow_gnn_amd has no CPU fallback.")
    import torch  # noqa: F401  (loads torch's libamdhip64 first)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sg_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {lib.sg_abi_version()}, this package needs {ABI_VERSION}: "
                          "rebuild it (python -c 'import __graft_entry__ as g; g.build(force=True)')")
    _lib = lib
    return lib


def check(code):
    if code == SG_OK:
        return
    msg = load().sg_last_error().decode(errors="replace")
    if code == SG_ERR_CAPACITY:
        raise CapacityError(code, msg)
    raise ShadowHipError(code, msg)
