"""shadow_gnn_amd -- MI355X-native hot path of shaDow-GNN.

Host-side mirror of the reference interfaces for the sampler + aggregation
path, over the C ABI of libshadow_hip.so (include/shadow_hip.h):

    sampler.py    ParallelSampler / SubgraphStructVec (reference pybind module)
                  and the device-resident fast path (HipSampler, DeviceBatch)
"""
import os as _os
import warnings as _warnings

__all__ = ["_lib", "sampler"]

# Environment switches of earlier rounds that became fixed settings: a script that still sets one would otherwise run a
# different configuration than it believes, silently.
_REMOVED_SWITCHES = ("SHADOW_BWD_AUX_STREAM", "SHADOW_CHAIN_SAGE_BWD", "SHADOW_DEBUG_TRACE", "SHADOW_DEFER_POINT", "SHADOW_FUSED_EPILOGUE",
                     "SHADOW_FUSED_LAYER_CALLS", "SHADOW_FUSE_GATHER_SPMM", "SHADOW_GEMM_SPLIT", "SHADOW_GEMM_TN_F16", "SHADOW_GRAD_PACK",
                     "SHADOW_MERGE_SUBGRAPHS", "SHADOW_ROOTS_SPARSE_GRAD", "SHADOW_ROW_STATS")
_set = [v for v in _REMOVED_SWITCHES if v in _os.environ]
if _set:
    _warnings.warn("shadow_gnn_amd: " + ", ".join(_set) + " no longer exist(s) -- the setting is fixed (the module attributes of "
                   "shadow_gnn_amd.ops / the sl_set_* entries are the A/B handles now) and the variable is ignored", RuntimeWarning, stacklevel=2)
del _os, _warnings, _set
