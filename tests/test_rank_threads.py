"""The in-process rank fabric of the GPU tests (tests/rank_threads.py) on the CPU: collectives inside row / column groups of
a 4x2 grid, point-to-point, gather, once(), and a failing rank that must not leave the others waiting."""
import time
import numpy as np
import pytest
from rank_threads import run_threads, ROW, COL


def test_group_collectives_on_a_4x2_grid():
    seen = {}

    def body(comm):
        r, c = comm.myrow, comm.mycol
        assert comm.rank == r + 4 * c                              # column-major grid ordering
        a = np.full(5, float(comm.rank))
        comm.allreduce(ROW, a)                                    # my grid row: ranks r and r + 4
        assert np.all(a == r + (r + 4))
        b = np.full(3, float(comm.rank))
        comm.allreduce(COL, b)                                    # my grid column: ranks 4c .. 4c+3
        assert np.all(b == sum(range(4 * c, 4 * c + 4)))
        d = np.full(4, float(comm.rank))
        comm.bcast(COL, d, 2)
        assert np.all(d == 2 + 4 * c)
        d = np.full(4, float(comm.rank))
        comm.bcast(ROW, d, 1)
        assert np.all(d == r + 4)
        # ring shift inside the column group, different lengths per rank
        send = np.full(10 + r, float(comm.rank))
        up, down = (r - 1) % 4, (r + 1) % 4
        recv = np.empty(10 + up)
        comm.sendrecv(COL, send, down, recv, up)
        assert np.all(recv == up + 4 * c)
        got = comm.all_gather_object((comm.rank, r, c))
        assert got == [(k, k % 4, k // 4) for k in range(8)]
        v = comm.once("token", lambda: object())
        seen[comm.rank] = v
        w = comm.once("token", lambda: object())                   # second use of a key: a new value, again shared
        assert w is not v

    run_threads(4, 2, body)
    assert len(seen) == 8 and len({id(v) for v in seen.values()}) == 1


def test_allreduce_results_are_identical_on_all_members():
    out = {}

    def body(comm):
        rng = np.random.default_rng(comm.rank)
        a = rng.standard_normal(1000)
        comm.allreduce(COL, a)
        out[comm.rank] = a.copy()

    run_threads(3, 2, body)
    for c in range(2):
        for r in range(1, 3):
            assert np.array_equal(out[3 * c], out[3 * c + r])


def test_a_failing_rank_releases_the_others():
    def body(comm):
        if comm.rank == 3:
            raise ValueError("rank 3 gives up")
        comm.allreduce(COL, np.zeros(4))
        comm.barrier()

    t = time.time()
    with pytest.raises(AssertionError, match="rank 3 gives up"):
        run_threads(2, 2, body)
    assert time.time() - t < 30
