"""world_size-2 / -4 gloo tests on CPU: the distributed oracle (reference pChASECPU kernels restated on gloo) built on
the product's layout helpers agrees with the serial oracle and with the reference's distributed HEMM known answer."""
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
