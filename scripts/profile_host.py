"""cProfile of the host side of the train step at a small batch (arxiv shape, SAGE-5, 256 roots; or gcn3 / 32 roots):
which Python frames the host-bound configurations spend their time in (development aid)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shadow_gnn_amd import dist as sdist
from shadow_gnn_amd.minibatch import TRAIN, MinibatchShallowExtractor
from shadow_gnn_amd.models import DeepGNN
from shadow_gnn_amd.optim import FlatAdam
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
kind = sys.argv[1] if len(sys.argv) > 1 else "sage"
dev = torch.device("cuda:0")
N, nnz, F0, C = SHAPES["arxiv"]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE["arxiv"])
g = torch.Generator(device=dev); g.manual_seed(1)
feat = torch.randn(N, F0, generator=g, device=dev); label = torch.randint(0, C, (N,), generator=g, device=dev)
B, L = (256, 5) if kind == "sage" else (32, 3)
steps = 300
roots = np.resize(np.random.default_rng(2).permutation(N), B * (steps + 20)).astype(np.int64)
mb = MinibatchShallowExtractor.on_device({TRAIN: (indptr, indices)}, {TRAIN: roots},
                                         dict(method="khop", depth=2, budget=20, add_self_edge=(kind != "sage")), ("hops",), feat, label,
                                         batch_size=B, device=dev, seed_cpp=3)
mb.lazy_features = True
mb.epoch_start_reset(0, TRAIN); mb.shuffle_entity(TRAIN, perm=np.arange(roots.size))
arch = dict(num_layers=L, num_cls_layers=1, heads=1, dim=256, act="elu", layer_norm="norm_feat", feature_augment_ops="sum",
            aggr=kind, residue="none", pooling="center", loss="softmax")
m = DeepGNN(F0, F0, C, 0, arch, [("hops", mb.get_aug_dim("hops"))], 1, dict(dropout=0.25, dropedge=0.15, lr=2e-5), "node").to(dev)
m.grad_sync = sdist.GradSync(m.parameters(), world_size=1)
m.optimizer = FlatAdam(m.grad_sync, lr=2e-5)
for _ in range(10):
    m.step(TRAIN, "running", mb.one_batch(TRAIN))
torch.cuda.synchronize()
pr = cProfile.Profile()
import time
torch.autograd.set_multithreading_enabled(False)      # (backward in this thread: its Python frames show up in the profile)
t0 = time.perf_counter()
pr.enable()
for _ in range(steps):
    m.step(TRAIN, "running", mb.one_batch(TRAIN))
pr.disable()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step under the profiler")
st = pstats.Stats(pr); st.sort_stats("tottime")
st.print_stats(45)
st.sort_stats("cumulative")
st.print_stats(70)
