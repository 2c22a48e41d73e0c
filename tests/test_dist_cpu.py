"""world_size-2 / -4 gloo tests on CPU: the distributed oracle (reference pChASECPU kernels restated on gloo, with its OWN
layout arithmetic) agrees with the serial oracle, with the reference's distributed HEMM known answer, and - index for index -
with the product's layout helpers."""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PORT = [29811]


@pytest.mark.parametrize("nranks,mb,typ", [(2, 0, "d"), (2, 16, "z"), (4, 0, "z"), (4, 8, "d")])
def test_distributed_oracle_on_gloo(nranks, mb, typ):
    _PORT[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_PORT[0]),
           os.path.join(ROOT, "tests", "dist_cpu_worker.py"), str(mb), typ]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0 and "DIST_CPU_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])


def test_block_layout_matches_reference_rule():
    from chase_amd import dist as cd
    # linalg/distMatrix/distMatrix.hpp:1992-2052: len = N/p (+1 if it does not divide), last rank takes the remainder
    for N, p in [(1001, 4), (10, 4), (4096, 2), (7, 7), (65536, 4)]:
        lay = cd.Layout(N, 0, p)
        ln = N // p if N % p == 0 else min(N, N // p + 1)
        assert lay.nb == ln
        counts = [lay.count(q) for q in range(p)]
        assert sum(counts) == N
        assert counts[:-1] == [ln] * (p - 1) or N < p * ln
    # numroc (distMatrix.hpp:44-67) against ScaLAPACK's definition
    lay = cd.Layout(1001, 64, 2)
    assert [lay.count(0), lay.count(1)] == [512, 489]
    assert cd.grid_shape(8) == (4, 2) and cd.grid_shape(4) == (2, 2) and cd.grid_shape(2) == (2, 1)


def test_grid_coordinates_and_shard_shapes_reference_known_answers():
    """tests/grid/mpiGrid2D.cpp:80-132 (ColMajor 2 x 2: rank -> (row, col) = 0:(0,0) 1:(1,0) 2:(0,1) 3:(1,1)) and
    tests/matrix/distMatrix.cpp:343-389 (11 x 11, 2 x 2 blocks block-cyclic on 2 x 2: local shapes 6x6, 5x6, 6x5, 5x5)."""
    from chase_amd import dist as cd
    assert [cd.coords_of(r, 2) for r in range(4)] == [(0, 0), (1, 0), (0, 1), (1, 1)]
    rl, cl = cd.Layout(11, 2, 2), cd.Layout(11, 2, 2)
    shapes = []
    for r in range(4):
        i, j = cd.coords_of(r, 2)
        shapes.append((rl.count(i), cl.count(j)))
    assert shapes == [(6, 6), (5, 6), (6, 5), (5, 5)]
    # local <-> global index maps are inverse to each other and partition 0..N-1 (block and block-cyclic)
    for (N, nb, p) in [(11, 2, 2), (1001, 64, 4), (10, 0, 4), (65536, 64, 4), (37, 5, 3)]:
        lay = cd.Layout(N, nb, p)
        seen = []
        for q in range(p):
            g = lay.globals_of(q)
            assert all(lay.owner(int(x)) == q for x in g[:50]) and all(lay.local(int(x)) == l for l, x in enumerate(g[:50]))
            seen.extend(int(x) for x in g)
        assert sorted(seen) == list(range(N))


def test_oracle_layout_is_independent_of_the_product_and_agrees_with_it():
    """oracle/dist_oracle.py restates the block rule and numroc itself (distMatrix.hpp:44-67,1992-2052); the product computes
    them in C (chase_hip_block_len / numroc / owner / local_index).  Two implementations, one rule: equal on the reference's
    known answers and on awkward sizes, zero-row ranks included (N = 9 on 4 ranks: 3, 3, 3, 0)."""
    from chase_amd import dist as cd
    from oracle.dist_oracle import OLayout, grid_coords
    import inspect
    import oracle.dist_oracle as D
    assert "chase_amd" not in inspect.getsource(D).split('"""', 2)[2]        # (outside its doc string: no import of the product)
    for (N, nb, p) in [(11, 2, 2), (1001, 64, 4), (10, 0, 4), (9, 0, 4), (65536, 64, 4), (37, 5, 3), (203, 16, 2), (7, 0, 7), (5, 0, 2)]:
        o, q = OLayout(N, nb, p), cd.Layout(N, nb, p)
        assert o.nb == q.nb
        assert [o.count(i) for i in range(p)] == [q.count(i) for i in range(p)]
        for g in range(0, N, max(1, N // 997)):
            assert o.owner(g) == q.owner(g) and o.local(g) == q.local(g)
    assert [OLayout(9, 0, 4).count(i) for i in range(4)] == [3, 3, 3, 0]
    assert [OLayout(11, 2, 2).count(i) for i in range(2)] == [6, 5]              # tests/matrix/distMatrix.cpp:343-389
    assert [grid_coords(r, 2) for r in range(4)] == [(0, 0), (1, 0), (0, 1), (1, 1)]   # tests/grid/mpiGrid2D.cpp:80-132
