"""shadow_gnn_amd -- MI355X-native hot path of shaDow-GNN.

Host-side mirror of the reference interfaces for the sampler + aggregation
path, over the C ABI of libshadow_hip.so (include/shadow_hip.h):

    sampler.py    ParallelSampler / SubgraphStructVec (reference pybind module)
                  and the device-resident fast path (HipSampler, DeviceBatch)
"""
__all__ = ["_lib", "sampler"]
