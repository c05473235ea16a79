"""The rank runner of the multi-process tests (tests/test_minibatch_gpu.py::_run_ranks) on plain CPU worker processes: results are
collected per rank, a rank process that dies fails the attempt at once (not after the queue's time-out), such an attempt is repeated
once, and a second death -- or anything else -- is an error."""
import os
import time

import pytest

from tests.test_minibatch_gpu import _run_ranks


def _ok_worker(rank, world, port, q, tag):
    q.put((rank, tag, world))


def _dies_once_worker(rank, world, port, q, flag):
    # rank 1 dies in the first attempt only (the flag file remembers it); the others would wait for it forever
    if rank == 1 and not os.path.exists(flag):
        open(flag, "w").close()
        os._exit(3)
    if not os.path.exists(flag):
        time.sleep(60)                      # (first attempt: stands for a rank stuck in a collective; the runner kills it)
    q.put((rank, "second try"))


def _always_dies_worker(rank, world, port, q):
    if rank == 0:
        os._exit(5)
    time.sleep(60)


def test_results_are_collected_per_rank():
    res = _run_ranks(_ok_worker, 3, "x")
    assert res == {r: ("x", 3) for r in range(3)}


def test_a_dead_rank_fails_fast_and_the_attempt_is_repeated_once(tmp_path):
    flag = str(tmp_path / "died")
    t0 = time.monotonic()
    with pytest.warns(UserWarning, match="one more attempt"):
        res = _run_ranks(_dies_once_worker, 3, flag)
    assert res == {r: ("second try",) for r in range(3)}
    assert time.monotonic() - t0 < 45          # (not the sleeping ranks' 60 s, let alone the queue's 600 s)


def test_a_rank_that_dies_twice_is_an_error():
    t0 = time.monotonic()
    with pytest.warns(UserWarning), pytest.raises(RuntimeError, match=r"died.*\(0, 5\)"):
        _run_ranks(_always_dies_worker, 2)
    assert time.monotonic() - t0 < 45
