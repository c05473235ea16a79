"""sg_ppr_push (ordered mode) against the oracle on the full products-shape graph: tables must be bit-identical."""
import sys, time, numpy as np, torch
from oracle import sampler_oracle as so
from shadow_gnn_amd.sampler import HipSampler
from shadow_gnn_amd.ppr import ppr_approximate_device
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
shape = sys.argv[1] if len(sys.argv) > 1 else "products"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
N, nnz, F, C = SHAPES[shape]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE[shape])
ip, ix = indptr.cpu().numpy().view(np.uint32), indices.cpu().numpy().view(np.uint32)
so.build()
targets = np.random.default_rng(0).permutation(N)[:T].astype(np.uint32)
hs = HipSampler(indptr, indices, device=dev, seed=3)
t0 = time.time(); gl, gn, gs = ppr_approximate_device(hs, targets, 200, 0.85, 1e-5); t1 = time.time()
ref = so.ppr_approximate(ip, ix, targets, k=200, alpha=0.85, epsilon=1e-5, num_threads=64); t2 = time.time()
ok = np.array_equal(gl, ref.len)
for i in range(T):
    L = int(gl[i])
    ok &= np.array_equal(gn[i, :L], ref.neigh[i, :L]) and np.array_equal(gs[i, :L].view(np.uint32), ref.score[i, :L].view(np.uint32))
print(f"{shape}: {T} targets, k=200 eps=1e-5: hip {t1 - t0:.2f} s, oracle {t2 - t1:.2f} s (64 threads), tables identical: {bool(ok)}")
