"""CPU, gloo, world_size 2 and 3: the data-parallel pieces that do not need a GPU -- the epoch plan
(every rank takes the same number of steps, ragged last global batch, a rank whose share is empty) and
the bucketed gradient all-reduce (SUM of loss-weighted gradients == the single-process gradient on the
whole global batch, also with unequal / empty shares and with the overlap hooks armed)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_model():
    return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ELU(), torch.nn.Linear(16, 16), torch.nn.ELU(),
                               torch.nn.Linear(16, 3))


def _worker(rank, world, port, q, shares, num_buckets):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from shadow_gnn_amd.dist import GradSync, broadcast_array, broadcast_parameters, init_from_env
    r, _l, w = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(1234 + rank)                 # different init per rank ...
    model = _toy_model()
    broadcast_parameters(model)                    # ... made identical here
    sync = GradSync(model.parameters(), num_buckets=num_buckets)
    g = torch.Generator().manual_seed(7)
    total = sum(shares)
    X = torch.randn(total, 8, generator=g)
    y = torch.randint(0, 3, (total,), generator=g)
    lo = sum(shares[:rank])
    xs, ys = X[lo:lo + shares[rank]], y[lo:lo + shares[rank]]
    errs = []
    for _step in range(2):                         # twice: zero() re-arms the hooks
        sync.zero()
        if shares[rank] > 0:                       # a rank with an empty share skips forward / backward ...
            (torch.nn.functional.cross_entropy(model(xs), ys) * (shares[rank] / total)).backward()
            assert sync._next == len(sync._slices)     # the hooks issued every slice during backward (overlap)
        sync.all_reduce()                          # ... but still enters every collective
        flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])      # (views of sync.flat, every tensor on its own 128-byte line)
        # single-process reference on the whole batch
        ref = _toy_model()
        ref.load_state_dict(model.state_dict())
        torch.nn.functional.cross_entropy(ref(X), y).backward()
        ref_flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
        errs.append(float((flat - ref_flat).abs().max()))
    # param.grad stayed views of the bucket
    views_ok = all(sync.flat.data_ptr() <= p.grad.data_ptr() < sync.flat.data_ptr() + 4 * sync.flat.numel()
                   for p in model.parameters())
    perm = broadcast_array(np.random.default_rng(100 + rank).permutation(50).astype(np.int64))
    q.put((rank, max(errs), views_ok, len(sync._slices), perm.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shares,num_buckets", [((6, 6), 1), ((6, 6), 2), ((5, 4, 3), 3), ((2, 2, 0), 2), ((1, 0, 0), 2)])
def test_grad_allreduce_equals_full_batch_gradient(shares, num_buckets):
    world = len(shares)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, shares, num_buckets)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, views_ok, nslices, perm in res:
        assert err < 1e-6 and views_ok, (rank, err)
        assert 1 <= nslices <= num_buckets
        assert perm == res[0][4]                   # every rank ends up with rank 0's permutation


@pytest.mark.parametrize("E,B,G", [(103, 16, 4), (101, 16, 8), (5, 16, 8), (64, 16, 4), (7, 3, 2), (1, 8, 3), (0, 8, 2)])
def test_epoch_plan_same_steps_on_every_rank_and_exact_global_batches(E, B, G):
    """Default plan: step t of all ranks together == global batch t of the permutation, in order; shares differ by
    at most one; every rank has ceil(E / B) steps (a share may be empty: E = 101, B = 16, G = 8 leaves a tail of 5)."""
    from shadow_gnn_amd.minibatch import plan_epoch
    order = np.random.default_rng(E + B + G).permutation(E)
    plans = [plan_epoch(order, B, G, r) for r in range(G)]
    T = -(-E // B)
    for mine, local, glob in plans:
        assert local.size == glob.size == T and mine.size == local.sum()
        assert np.array_equal(glob, plans[0][2])
    assert np.array_equal(np.sort(np.concatenate([p[0] for p in plans])), np.sort(order))
    cursors = [0] * G
    for t in range(T):
        parts = []
        for r, (mine, local, glob) in enumerate(plans):
            parts.append(mine[cursors[r]:cursors[r] + local[t]])
            cursors[r] += local[t]
        got = np.concatenate(parts)
        assert np.array_equal(got, order[t * B:(t + 1) * B])
        sizes = [p[1][t] for p in plans]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == plans[0][2][t]
    if (E, B, G) == (101, 16, 8):
        assert [int(p[1][-1]) for p in plans] == [1, 1, 1, 1, 1, 0, 0, 0]      # three ranks idle in the last step


@pytest.mark.parametrize("E,B,G", [(103, 16, 4), (70, 16, 3), (5, 16, 8), (9, 4, 2)])
def test_epoch_plan_static_partition_keeps_roots_on_their_rank(E, B, G):
    """Cache mode: position p of the entity set always lands on rank p % G, whatever the permutation; the
    per-step sizes are known to every rank (their sum is the step's global batch)."""
    from shadow_gnn_amd.minibatch import plan_epoch
    T = -(-E // B)
    for seed in (0, 1):
        order = np.random.default_rng(seed).permutation(E)
        plans = [plan_epoch(order, B, G, r, static_partition=True) for r in range(G)]
        for r, (mine, local, glob) in enumerate(plans):
            assert np.all(mine % G == r) and mine.size == local.sum() == np.sum(np.arange(E) % G == r)
            assert local.size == glob.size == T
            assert np.array_equal(mine, order[order % G == r])               # epoch order preserved inside the share
            assert local.max() - local.min() <= 1
        assert np.array_equal(sum(p[1] for p in plans), plans[0][2]) and plans[0][2].sum() == E


def test_deferred_work_registry():
    """ops.defer / ops.fire_deferred: one pending callable per key (a newer one replaces an unfired older one), fired
    at the configured point or, whatever is left, at "fwd"."""
    from shadow_gnn_amd import ops
    ops._DEFERRED.clear()
    fired = []
    ops.defer(("a", 0), lambda: fired.append("old"))
    ops.defer(("a", 0), lambda: fired.append("new"))
    ops.defer(("b", 0), lambda: fired.append("other"))
    other = "agg" if ops.DEFER_POINT != "agg" else "body"
    ops.fire_deferred(other)
    assert fired == []
    ops.fire_deferred(ops.DEFER_POINT)
    assert sorted(fired) == ["new", "other"] and not ops._DEFERRED
    ops.defer(("a", 0), lambda: fired.append("late"))
    ops.fire_deferred("fwd")
    assert fired[-1] == "late" and not ops._DEFERRED


def test_flat_adam_steps_without_an_explicit_all_reduce_and_speaks_torch_adam_state():
    """ GradSync packs gradients lazily; FlatAdam must gather them itself when nobody called
    GradSync.all_reduce -- otherwise it silently updates from an all-zero buffer.  Three steps equal torch.optim.Adam's,
    and the optimizer state round-trips through torch.optim.Adam's state_dict layout in both directions."""
    import copy
    from shadow_gnn_amd.dist import GradSync
    from shadow_gnn_amd.optim import FlatAdam
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ref = copy.deepcopy(net)
    x, y = torch.randn(16, 6), torch.randn(16, 3)
    sync = GradSync(net.parameters(), world_size=1)
    opt = FlatAdam(sync, lr=0.01)
    topt = torch.optim.Adam(ref.parameters(), lr=0.01)
    for _ in range(3):
        sync.zero()
        ((net(x) - y) ** 2).mean().backward()
        opt.step()                                           # no sync.all_reduce() in between
        topt.zero_grad()
        ((ref(x) - y) ** 2).mean().backward()
        topt.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (a - b).abs().max()
    # FlatAdam -> torch Adam
    t2 = torch.optim.Adam(ref.parameters(), lr=0.5)
    t2.load_state_dict(opt.state_dict())
    assert t2.param_groups[0]["lr"] == 0.01
    for i, p in enumerate(ref.parameters()):
        assert torch.allclose(t2.state[p]["exp_avg"], topt.state[p]["exp_avg"], rtol=1e-5, atol=1e-8)
    # torch Adam -> FlatAdam
    opt2 = FlatAdam(GradSync(copy.deepcopy(net).parameters(), world_size=1), lr=0.5)
    opt2.load_state_dict(topt.state_dict())
    assert opt2.step_count == 3 and opt2.lr == 0.01
    assert torch.allclose(opt2.exp_avg, opt.exp_avg, rtol=1e-5, atol=1e-8) and torch.allclose(opt2.exp_avg_sq, opt.exp_avg_sq, rtol=1e-5, atol=1e-10)
    # round 2's flat layout (moments packed back to back, no line padding) lands at the padded offsets
    packed = dict(step=3, lr=0.01, betas=(0.9, 0.999), eps=1e-8,
                  exp_avg=torch.cat([topt.state[p]["exp_avg"].reshape(-1) for p in ref.parameters()]),
                  exp_avg_sq=torch.cat([topt.state[p]["exp_avg_sq"].reshape(-1) for p in ref.parameters()]))
    opt3 = FlatAdam(GradSync(copy.deepcopy(net).parameters(), world_size=1), lr=0.5)
    assert opt3.exp_avg.numel() > packed["exp_avg"].numel()          # (the padded buffers are longer: a plain copy_ would raise)
    opt3.load_state_dict(packed)
    assert opt3.step_count == 3 and torch.allclose(opt3.exp_avg, opt.exp_avg, rtol=1e-5, atol=1e-8)
    assert torch.allclose(opt3.exp_avg_sq, opt.exp_avg_sq, rtol=1e-5, atol=1e-10)
    with pytest.raises(ValueError):
        opt3.load_state_dict(dict(packed, exp_avg=packed["exp_avg"][:-1]))


def _pin_worker(local_rank, local_world, q):
    import os
    from shadow_gnn_amd import dist as sdist
    before = sorted(os.sched_getaffinity(0))
    info = sdist.pin_host_threads(local_rank, local_world)
    q.put((local_rank, before, sorted(os.sched_getaffinity(0)), info, torch.get_num_threads()))


def test_pin_host_threads_gives_every_rank_its_own_cpu_slice():
    """dist.pin_host_threads: N ranks on one host get disjoint, contiguous slices of the allowed hardware threads that
    together cover them (up to the remainder), torch's intra-op pool is capped to the slice, and a single-rank run is left
    alone (bench.py's CPU baselines want every core)."""
    import os
    import torch.multiprocessing as mp
    from shadow_gnn_amd import dist as sdist
    avail = sorted(os.sched_getaffinity(0))
    assert sdist.pin_host_threads(0, 1) == dict(pinned=False) and sorted(os.sched_getaffinity(0)) == avail
    if len(avail) < 2:
        pytest.skip("one hardware thread: nothing to slice")
    world = 2 if len(avail) < 8 else 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pin_worker, args=(r, world, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per = len(avail) // world
    seen = []
    for r, before, after, info, nthreads in res:
        assert before == avail and info["pinned"] and after == avail[r * per:(r + 1) * per]
        assert info["n_cpus"] == per and nthreads == min(8, per) == info["torch_threads"]
        seen += after
    assert len(seen) == len(set(seen)) == per * world
