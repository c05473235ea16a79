"""Several rank processes sharing ONE GPU over gloo: the data-parallel paths (epoch plan with ragged / empty shares, GradSync's
bucket all-reduces from the backward hooks, per-rank subgraph caches, bench.py under torch.distributed.run) with 2, 3 and 8 ranks.
Kept in a file of its own that sorts LAST: processes time-slicing a device is an arrangement the product does not run in (one process
per GPU); it is here as functional evidence, and a device-level fault of that arrangement must not hide the rest of the suite from
`pytest -x`.  Workers and the rank runner live in tests/test_minibatch_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch

from tests.test_minibatch_gpu import DEV, _dp_worker, _ragged_worker, _run_ranks, _setup, _single_process_ragged_epoch

pytestmark = pytest.mark.gpu


def test_two_rank_data_parallel_step_equals_single_process():
    from shadow_gnn_amd.minibatch import TRAIN
    from shadow_gnn_amd.models import DeepGNN
    res = {r: v[0] for r, v in _run_ranks(_dp_worker, 2).items()}
    # both ranks end with identical parameters
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
    # single process, global batch 16, same initial parameters (rank 0's seed), deterministic
    # (full 2-hop) sampler -> the same three optimizer steps up to fp32 summation order
    mb = _setup(prefetch=False, batch=16, aug=(), budget=-1)[0]
    torch.manual_seed(5)
    arch = dict(num_layers=2, heads=1, dim=32, act="elu", aggr="sage", residue="none", pooling="center")
    model = DeepGNN(20, 20, 7, 0, arch, [], 1, dict(dropout=0.0, dropedge=0.0, lr=1e-2), "node").to(DEV)
    model.optimizer = torch.optim.Adam(model.parameters(), lr=1e-2)
    for _ in range(3):
        model.step(TRAIN, "running", mb.one_batch(TRAIN))
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(res[0][k], v.cpu().numpy(), rtol=2e-3, atol=2e-3, err_msg=k)


@pytest.mark.parametrize("nroots", [103, 97])
def test_three_rank_ragged_epoch_equals_single_process(nroots):
    """A whole epoch whose last global batch is ragged (7 roots over 3 ranks) or leaves two ranks EMPTY (1 root):
    nobody hangs in the all-reduce, every rank takes ceil(E / B) steps, all ranks end with identical parameters, and
    those equal the single-process run over the same global batches (loss-weighted SUM all-reduce)."""
    from shadow_gnn_amd.minibatch import TRAIN, MinibatchShallowExtractor
    from shadow_gnn_amd.models import DeepGNN
    from shadow_gnn_amd.synthetic import make_graph_numpy
    res = _run_ranks(_ragged_worker, 3, nroots, 0)
    T = -(-nroots // 16)
    tail = nroots - 16 * (T - 1)
    for r in range(3):
        sizes = res[r][0]
        assert len(sizes) == T
        assert [s for s, _w in sizes[:-1]] == [6 - (r > 0)] * (T - 1)            # 16 = 6 + 5 + 5
        assert sizes[-1][0] == tail // 3 + (r < tail % 3)
        assert abs(sizes[-1][1] - sizes[-1][0] / tail) < 1e-6
    if nroots == 97:
        assert [res[r][0][-1][0] for r in range(3)] == [1, 0, 0]
    for k in res[0][2]:
        assert np.array_equal(res[0][2][k], res[1][2][k]) and np.array_equal(res[0][2][k], res[2][2][k]), k
    single = _single_process_ragged_epoch(nroots)
    for k, v in single.items():
        np.testing.assert_allclose(res[0][2][k], v, rtol=2e-3, atol=2e-3, err_msg=k)


def test_two_rank_ppr_cache_survives_reshuffled_epochs():
    """Per-rank record -> reuse caches with a NEW permutation every epoch: the static root -> rank map keeps
    every reused root on the rank that recorded it; three epochs, no 'never recorded' error, ranks stay in step."""
    res = _run_ranks(_ragged_worker, 2, 70, 3)
    for r in range(2):
        sizes, modes, _sd = res[r]
        assert modes == ["record", "reuse", "reuse"]
        assert len(sizes) == 3 * 5 and sum(s for s, _w in sizes) == 3 * 35
    for k in res[0][2]:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k


def test_bench_two_ranks_through_torch_distributed_run():
    """ bench.py exactly as the driver launches it for N > 1 -- `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ... bench.py --gpus 2` -- on the smallest workload, the two ranks
    sharing this box's one GPU over gloo (SHADOW_DIST_BACKEND; RCCL refuses two ranks on one device): rank 0 prints ONE
    JSON line that carries the contract's fields for a 2-rank weak-scaling run, and the whole-job rate is that of two
    batches per step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SHADOW_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "3", "--warmup", "1", "--workload", "arxiv-khop-gcn3", "--no-cpu-baseline", "--no-tail"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, two.stdout[-2000:]                 # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 64
    assert d["metric"] == "sampled-nodes/sec" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "roofline" in d and "host_busy_ms_per_step" in d
    # two batches of 32 roots per step: about twice the nodes of one
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, cwd=root, env=env,
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    ratio = d["config"]["nodes_per_step"] / d1["config"]["nodes_per_step"]
    assert 1.6 < ratio < 2.4, ratio


def test_eight_rank_ragged_epoch_on_one_gpu_equals_single_process():
    """EIGHT ranks -- the node's process count -- sharing cuda:0 over gloo: 103 roots in global batches of 16 (2 per rank), the
    last batch 7 roots (ranks 0..6 one root, rank 7 an EMPTY share).  Every rank issues the same sequence of bucket all-reduces
    through GradSync's backward hooks (nobody hangs), all eight end with identical parameters, and those equal the single-process
    run over the same global batches.  Functional evidence only -- eight processes time-slicing one GPU say nothing about scaling
    (profiles/r06_dist_8proc_one_gpu.json holds the host-side figures of the same arrangement).
    (A rank process that DIES gets the arrangement one more try, see _run_ranks; wrong sizes or parameters never do.)"""
    nroots = 103
    res = _run_ranks(_ragged_worker, 8, nroots, 0)
    T = -(-nroots // 16)
    for r in range(8):
        sizes = res[r][0]
        assert len(sizes) == T
        assert [s for s, _w in sizes[:-1]] == [2] * (T - 1)
        assert sizes[-1][0] == (1 if r < 7 else 0)
        assert abs(sizes[-1][1] - sizes[-1][0] / 7) < 1e-6
    for k in res[0][2]:
        for r in range(1, 8):
            assert np.array_equal(res[0][2][k], res[r][2][k]), (k, r)
    single = _single_process_ragged_epoch(nroots)
    for k, v in single.items():
        np.testing.assert_allclose(res[0][2][k], v, rtol=2e-3, atol=2e-3, err_msg=k)
