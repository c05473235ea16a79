import numpy as np, torch, sys
from shadow_gnn_amd.minibatch import TRAIN, MinibatchShallowExtractor
from shadow_gnn_amd.models import DeepGNN
from shadow_gnn_amd.synthetic import make_graph_torch
dev = torch.device("cuda:0")
N, nnz, F0 = 400000, 12000000, 100
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=5000)
g = torch.Generator(device=dev); g.manual_seed(1)
feat = torch.randn(N, F0, generator=g, device=dev)
import os
AGGR = os.environ.get("AGGR", "sage"); RES = os.environ.get("RES", "none"); POOL = os.environ.get("POOL", "center")
def run(C, prefetch, steps=int(os.environ.get("STEPS", "40")), B=int(os.environ.get("B", "1024"))):
    g2 = torch.Generator(device=dev); g2.manual_seed(2)
    label = torch.randint(0, C, (N,), generator=g2, device=dev)
    roots = np.random.default_rng(2).permutation(N)[:B * (steps + 2)].astype(np.int64)
    mb = MinibatchShallowExtractor.on_device({TRAIN: (indptr, indices)}, {TRAIN: roots}, dict(method="khop", depth=2, budget=20, add_self_edge=(AGGR != "sage")), (), feat, label,
                                   batch_size=B, device=dev, seed_cpp=3, prefetch=prefetch)
    mb.epoch_start_reset(0, TRAIN); mb.shuffle_entity(TRAIN, perm=np.arange(roots.size))
    torch.manual_seed(4)
    arch = dict(num_layers=5, num_cls_layers=1, heads=4 if AGGR == "gat" else 1, dim=256, act="relu", layer_norm="norm_feat", feature_augment_ops="sum", aggr=AGGR, residue=RES, pooling=POOL, loss="softmax")
    m = DeepGNN(F0, F0, C, 0, arch, [], 1, dict(dropout=0.4, dropedge=0.05, lr=0.002), "node").to(dev)
    losses = []
    for _ in range(steps):
        losses.append(m.step(TRAIN, "running", mb.one_batch(TRAIN))["loss"].detach())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu().numpy(), torch.cat([p.detach().flatten() for p in m.parameters()]).cpu().numpy()
for C in (47,):
    for prefetch in (False, True):
        a = run(C, prefetch); b = run(C, prefetch)
        first = next((i for i in range(len(a[0])) if a[0][i] != b[0][i]), None)
        print(f"C={C} prefetch={prefetch}: losses identical {np.array_equal(a[0], b[0])} (first differing step {first}), params identical {np.array_equal(a[1], b[1])}")
    a = run(C, False); b = run(C, True)
    first = next((i for i in range(len(a[0])) if a[0][i] != b[0][i]), None)
    print(f"C={C} prefetch False vs True: losses identical {np.array_equal(a[0], b[0])} (first differing step {first})")
