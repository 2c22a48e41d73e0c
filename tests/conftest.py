import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_FIX = os.path.join(GOLDEN, "ref_fixtures")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_ref_matrix(name, m, n, cplx):
    """Reader of the reference's raw column-major binary fixtures (tests/linalg/internal/utils.hpp:113-135)."""
    dt = np.complex128 if cplx else np.float64
    a = np.fromfile(os.path.join(REF_FIX, name), dtype=dt)
    assert a.size == m * n, (name, a.size, m, n)
    return np.asfortranarray(a.reshape((m, n), order="F"))


@pytest.fixture(scope="session")
def ctx():
    from chase_amd.capi import Context
    c = Context(0)
    yield c
    c.close()
