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


# GPU files in this order: kernel-level parity first, whole solves, pseudo-Hermitian, the reference's unit tests, the grid
# Impl with ranks as threads - all inside this one process - and only then the tests that start other processes (ranks as
# processes, bench.py), so that nothing a process-count limit of the box does to those can hide the parity evidence
_ORDER = ["test_gpu_kernels", "test_gpu_solve", "test_gpu_pseudo", "test_gpu_reference_unit_tests", "test_gpu_dist",
          "test_gpu_replay", "test_gpu_fullsize", "test_gpu_processes", "test_gpu_bench"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else -1          # CPU files keep their place in front
    items.sort(key=key)                                                # stable: order inside a file is untouched


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
