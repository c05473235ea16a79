"""CPU, world_size 2, gloo: the data-parallel pieces that do not need a GPU --
root sharding of the shared permutation and the single-bucket gradient
all-reduce (equivalence with the single-process gradient on the full batch)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from shadow_gnn_amd.dist import GradSync, broadcast_parameters, init_from_env
    r, _l, w = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(1234 + rank)                 # different init per rank ...
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ELU(), torch.nn.Linear(16, 3))
    broadcast_parameters(model)                    # ... made identical here
    sync = GradSync(model.parameters())
    g = torch.Generator().manual_seed(7)
    X = torch.randn(12, 8, generator=g)
    y = torch.randint(0, 3, (12,), generator=g)
    B = 12 // world
    xs, ys = X[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B]
    sync.zero()
    torch.nn.functional.cross_entropy(model(xs), ys).backward()
    sync.all_reduce()
    flat = sync.flat.clone()
    # single-process reference on the whole batch
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ELU(), torch.nn.Linear(16, 3))
    ref.load_state_dict(model.state_dict())
    torch.nn.functional.cross_entropy(ref(X), y).backward()
    ref_flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    # param.grad stayed views of the bucket
    views_ok = all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in model.parameters())
    q.put((rank, float((flat - ref_flat).abs().max()), views_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_equals_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, views_ok in res:
        assert err < 1e-6 and views_ok, (rank, err)


def test_root_sharding_partitions_every_global_batch():
    """Every global batch of B roots is split into disjoint equal rank slices, in order."""
    ent = np.arange(1000, 1103)
    perm = np.random.default_rng(0).permutation(ent.size)
    B, G = 16, 4
    slices = []
    for r in range(G):
        e = ent[perm]
        nfull = (e.size // B) * B
        body = e[:nfull].reshape(-1, G, B // G)[:, r, :].reshape(-1)
        tail = e[nfull:]
        per = -(-tail.size // G)
        slices.append(np.concatenate([body, tail[r * per:(r + 1) * per]]))
    allr = np.concatenate(slices)
    assert np.array_equal(np.sort(allr), np.sort(ent))
    # step t of every rank together == global batch t of the permutation
    e = ent[perm]
    for t in range(e.size // B):
        got = np.concatenate([s[t * (B // G):(t + 1) * (B // G)] for s in slices])
        assert np.array_equal(got, e[t * B:(t + 1) * B])
