import numpy as np, torch, sys
from shadow_gnn_amd.sampler import HipSampler
from shadow_gnn_amd.ppr import ppr_approximate_device
from shadow_gnn_amd.synthetic import make_graph_numpy
from oracle import sampler_oracle as so
indptr, indices = make_graph_numpy(2000, 8, seed=5)
targets = np.arange(0, 256, dtype=np.uint32)
print("row0", indices[indptr[0]:indptr[1]], flush=True)
hs = HipSampler(indptr, indices, device=torch.device("cuda:0"), seed=0)
gl, gn, gs = ppr_approximate_device(hs, targets, 20, 0.85, 1e-4, hash_slots=1 << 12, num_waves=64)
ref = so.ppr_approximate(indptr, indices, targets, k=20, alpha=0.85, epsilon=1e-4)
print("len", gl, ref.len, "nb eq", np.array_equal(gn, ref.neigh), "sc eq", np.array_equal(gs.view(np.uint32), ref.score.view(np.uint32)))
d = np.abs(gs.astype(np.float64) - ref.score.astype(np.float64))
i, j = np.unravel_index(np.argmax(d), d.shape)
print("max abs diff", d.max(), "at", i, j, gs[i, j], ref.score[i, j], "rel", d.max() / ref.score[i, j])
print("n differing", (gs.view(np.uint32) != ref.score.view(np.uint32)).sum(), "of", gs.size)
print("first rows", gs[0, :5], ref.score[0, :5])
