"""Worker of tests/test_dist_rccl_gpu.py: ONE rank, backend nccl (= RCCL), SHADOW_DIST_FORCE_INIT=1 -- communicator
set-up on cuda:0, GPU-side broadcast of the epoch permutation and the parameters, two DeepGNN.step calls whose bucket
all-reduces are issued asynchronously from the post-accumulate hooks, FlatAdam behind them.  The same two steps are then
repeated by a second model without any process-group traffic (GradSync(world_size=1) before the group exists is not
possible in one process, so the comparison model uses overlap off + collective off).  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as tdist


def main():
    from shadow_gnn_amd import dist as sdist
    from shadow_gnn_amd.minibatch import TRAIN, MinibatchShallowExtractor
    from shadow_gnn_amd.models import DeepGNN
    from shadow_gnn_amd.optim import FlatAdam
    from shadow_gnn_amd.synthetic import make_graph_torch
    assert sdist.force_init()
    rank, local, world = sdist.init_from_env()
    assert world == 1 and tdist.is_initialized() and tdist.get_backend() == "nccl", (world, tdist.is_initialized())
    dev = torch.device("cuda", 0)
    N, F0, C, B = 20000, 64, 11, 48
    indptr, indices = make_graph_torch(N, N * 12, seed=0, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(1)
    feat = torch.randn(N, F0, generator=g, device=dev)
    lab = torch.randint(0, C, (N,), generator=g, device=dev)
    roots = np.arange(0, N, 7).astype(np.int64)
    arch = dict(num_layers=3, num_cls_layers=1, heads=1, dim=128, act="relu", layer_norm="norm_feat", feature_augment_ops="sum",
                aggr="sage", residue="none", pooling="center", loss="softmax")
    out = {}

    def run(collective):
        mb = MinibatchShallowExtractor.on_device({TRAIN: (indptr, indices)}, {TRAIN: roots}, dict(method="khop", depth=2, budget=10,
                                                 add_self_edge=False), (), feat, lab, batch_size=B, device=dev, seed_cpp=3, rank=0,
                                                 world_size=1, prefetch=False)
        mb.epoch_start_reset(0, TRAIN)
        np.random.seed(5)
        mb.shuffle_entity(TRAIN)                       # perm=None: drawn here and (collective) broadcast on the GPU through RCCL
        torch.manual_seed(4)
        model = DeepGNN(F0, F0, C, 0, arch, [], 1, dict(dropout=0.0, dropedge=0.0, lr=0.01), "node").to(dev)
        if collective:
            sdist.broadcast_parameters(model)
        gs = sdist.GradSync(model.parameters(), world_size=1)
        if not collective:
            gs.collective = gs.overlap = False
        assert gs.collective == collective and gs.overlap == collective
        model.grad_sync = gs
        model.optimizer = FlatAdam(gs, lr=0.01)
        losses = []
        for _ in range(2):
            ret = model.step(TRAIN, "running", mb.one_batch(TRAIN))
            losses.append(float(ret["loss"]))
        torch.cuda.synchronize()
        return losses, gs, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu(), mb.entity_epoch[TRAIN].copy()

    l1, gs1, p1, e1 = run(True)
    l0, gs0, p0, e0 = run(False)
    # an explicit GPU-side broadcast_array
    arr = sdist.broadcast_array(np.arange(17, dtype=np.int64) * 3, src=0, device=dev)
    tdist.barrier()
    out = dict(losses_rccl=l1, losses_plain=l0, issued=gs1.issued, issued_plain=gs0.issued, buckets=len(gs1._slices),
               max_param_diff=float((p1 - p0).abs().max()), same_epoch=bool(np.array_equal(e1, e0)),
               bcast_ok=bool(np.array_equal(arr, np.arange(17) * 3)), backend=tdist.get_backend(),
               allreduce_host_wait_ms=gs1.wait_s * 1e3)
    print(json.dumps(out), flush=True)
    tdist.destroy_process_group()


if __name__ == "__main__":
    main()
