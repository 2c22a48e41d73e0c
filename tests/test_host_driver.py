"""Host logic without a GPU: the product's C++ solver driver (chase_amd/host/algorithm.hpp — the restatement of ChASE's
algorithm.inc: solve, filter, calc_degrees, locking, lanczos + DoS bounds) is compiled with g++ against a naive CPU mock of
the ChaseBase surface (tests/host_driver_harness.cpp, test infrastructure) and compared with the Python oracle on the same
matrix: iteration count, filtered-vector count, the whole sequence of virtual calls with their scalar arguments (the
"golden call trace" of SURVEY.md §8 row A0) and the eigenvalues."""
import os
import re
import shutil
import subprocess
import numpy as np
import pytest
from oracle import chase_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_NUM = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path_factory.mktemp("hd") / "host_driver_harness"
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", str(exe), os.path.join(ROOT, "tests", "host_driver_harness.cpp")],
                   check=True, cwd=ROOT)
    return str(exe)


def _same_trace_line(a, b):
    ta, tb = a.split(), b.split()
    if len(ta) != len(tb):
        return False
    for x, y in zip(ta, tb):
        if _NUM.match(x) and _NUM.match(y):
            fx, fy = float(x), float(y)
            if abs(fx - fy) > 1e-7 * max(1.0, abs(fx), abs(fy)):
                return False
        elif x != y:
            return False
    return True


@pytest.mark.parametrize("N,nev,nex,deg,opt", [(96, 8, 6, 10, 1), (120, 12, 8, 20, 0), (150, 10, 10, 16, 1)])
def test_cpp_driver_issues_the_oracles_call_sequence(harness, N, nev, nex, deg, opt):
    out = subprocess.run([harness, str(N), str(nev), str(nex), str(deg), str(opt)], check=True, capture_output=True,
                         text=True, timeout=600).stdout.splitlines()
    got = {k: int(v) for k, v in (l.split() for l in out if l.split()[0] in ("iterations", "filtered_vecs", "locked"))}
    lam = np.array([float(l.split()[1]) for l in out if l.startswith("lambda ")])
    res = np.array([float(l.split()[2]) for l in out if l.startswith("lambda ")])
    trace = [l[len("trace "):] for l in out if l.startswith("trace ")]

    k = O.OracleCPU(O.clement(N, False, perturb=0), nev, nex)
    k.config.deg, k.config.opt = deg, bool(opt)
    tr = []
    so = O.solve(k, tr)
    assert got["iterations"] == so["iterations"]
    assert got["filtered_vecs"] == so["filtered_vecs"]
    assert got["locked"] >= nev
    assert np.max(np.abs(lam - k.ritzv[:nev])) < 1e-9
    assert np.max(np.abs(lam - (-N + 2.0 * np.arange(nev)))) < 1e-9          # analytic Clement spectrum
    assert np.max(res) < 1e-10
    assert len(trace) == len(tr), (len(trace), len(tr))
    for a, b in zip(trace, tr):
        assert _same_trace_line(a, b), (a, b)
