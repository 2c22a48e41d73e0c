"""The C-ABI library loads on a CPU-only box and exports every symbol the public headers declare."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(chase_hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "chase_amd", "lib", "libchase_hip.so"))
    names = _declared("chase_hip.h") + _declared("chase_hip_solver.h") + _declared("chase_hip_grid.h")
    # every entry point of the application-facing C interface (sequential, distributed grid-handle forms, helpers)
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "chase_c_interface.h")).read(), flags=re.S)
    cnames = sorted(set(re.findall(r"\b(p?[dz]?chase_[A-Za-z0-9_]*)\s*\(", txt)))
    assert len(cnames) >= 30 and "pzchase_init_blockcyclic_hip_" in cnames and "dchase_" in cnames
    names += cnames
    assert len(names) > 80
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_mpi_front_end_exports_the_reference_signatures():
    """libchase_hip_mpi.so (built when mpi.h is present): the reference's own names for the distributed inits
    (interface/chase_c_interface.h:61-65,95-99,126-128,149)"""
    import pytest
    path = os.path.join(ROOT, "chase_amd", "lib", "libchase_hip_mpi.so")
    if not os.path.exists(path):
        pytest.skip("no MPI on this box: front end not built")
    out = os.popen(f"nm -D --defined-only {path}").read()
    # every entry point include/chase_c_interface_mpi.h declares (the front end includes that header: a signature that differs
    # from its declaration does not compile)
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "chase_c_interface_mpi.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(p[dz]chase_init[a-z_]*)\s*\(", txt)))
    assert len(declared) == 24, declared
    for n in declared:
        assert f" T {n}\n" in out, n
    assert '#include "../../include/chase_c_interface_mpi.h"' in open(os.path.join(ROOT, "chase_amd", "host", "c_interface_mpi.c")).read()
    for n in ("pdchase_init_", "pdchase_init_internal_", "pzchase_init_", "pzchase_init_internal_", "pzchase_init_pseudo_",
              "pdchase_init_blockcyclic_", "pdchase_init_blockcyclic_internal_", "pzchase_init_blockcyclic_",
              "pzchase_init_blockcyclic_internal_", "pzchase_init_pseudo_blockcyclic_"):
        assert f" T {n}\n" in out, n
        assert f" T {n[:-1]}_f_\n" in out, n + " (MPI_Fint twin, chase_c_interface.cpp:2425-3030)"


def test_fails_loudly_without_gpu_or_runs_on_one():
    from chase_amd import capi
    h = ctypes.c_void_p()
    rc = capi.lib.chase_hip_ctx_create(ctypes.byref(h), 0, None)
    if rc != 0:
        assert rc == -1002                      # CHASE_HIP_ENODEV: no silent CPU fallback
        assert b"HIP device" in capi.lib.chase_hip_last_error()
    else:
        capi.lib.chase_hip_ctx_destroy(h)


def test_host_lapack_provider_is_bound():
    from chase_amd import capi
    assert capi.lib.chase_hip_lapack_provider() != b""


def test_gemm_workspace_covers_the_rims_of_a_split_three_multiplication_product():
    """A complex filter product of arbitrary size is cut into a 3M bulk (whole 128-row and 8-deep tiles) and two thin 4M rims
    (gemm_mfma_f64.hip launch_gemm); the launcher REFUSES a workspace smaller than a piece's plan (it never re-plans to fit), so
    the size the context allocates for the whole shape must cover every piece.  Host-only function: no GPU needed."""
    from chase_amd.capi import lib
    need = lambda op, m, n, k, r=0: lib.chase_hip_gemm_workspace_bytes(1, op, m, n, k, 256, r)
    for op in (b"N", b"C"):
        for (m, n, k) in [(1001, 160, 1001), (8193, 640, 16384), (16384, 133, 8190), (65536, 2560, 65536), (1200, 140, 1200)]:
            m1, k1 = m - m % 128, k - k % 8
            whole = need(op, m, n, k)
            pieces = [need(op, m1, n, k1)]
            if k1 < k:
                pieces.append(need(op, m1, n, k - k1))
            if m1 < m:
                pieces.append(need(op, m - m1, n, k))
            assert whole >= max(pieces), (op, m, n, k, whole, pieces)
            assert need(op, m, n, k, 4) >= whole or need(op, m, n, k, 4) > 0       # shared-chip granularity: its own plan
    assert need(b"N", 0, 5, 5) == 0
    # round 6: the operand-sum plane of the 3M kernels (one 4 KB block per 64-column tile and K step) is part of the figure
    assert need(b"N", 65536, 2560, 65536) >= 40 * 8192 * 4096
    assert lib.chase_hip_gemm_workspace_bytes(0, b"N", 65536, 2560, 65536, 256, 0) < 40 * 8192 * 4096      # real: no plane


def test_stemr_rejects_non_finite_input_instead_of_hanging():
    """a Lanczos recurrence that overflowed hands NaN / inf to the tridiagonal eigensolver; LAPACK's MRRR need not return on
    such input (found in round 5: a replayed rank hung there) - the entry point refuses it.  Host-only: no GPU needed."""
    import numpy as np
    from chase_amd.capi import lib
    n = 6
    for bad in (np.nan, np.inf):
        d, e = np.arange(n, dtype=np.float64), np.full(n, 0.5)
        w, Z = np.zeros(n), np.zeros(n * n)
        e[2] = bad
        rc = lib.chase_hip_stemr_host(n, d.ctypes.data_as(ctypes.c_void_p), e.ctypes.data_as(ctypes.c_void_p),
                                      w.ctypes.data_as(ctypes.c_void_p), Z.ctypes.data_as(ctypes.c_void_p), n)
        assert rc != 0 and b"non-finite" in lib.chase_hip_last_error()
    d, e = np.arange(n, dtype=np.float64), np.full(n, 0.5)
    w, Z = np.zeros(n), np.zeros(n * n)
    e[n - 1] = np.nan                                    # the entry past the last off-diagonal is not read
    rc = lib.chase_hip_stemr_host(n, d.ctypes.data_as(ctypes.c_void_p), e.ctypes.data_as(ctypes.c_void_p),
                                  w.ctypes.data_as(ctypes.c_void_p), Z.ctypes.data_as(ctypes.c_void_p), n)
    T = np.diag(np.arange(n, dtype=np.float64)) + np.diag(np.full(n - 1, 0.5), 1) + np.diag(np.full(n - 1, 0.5), -1)
    assert rc == 0 and np.allclose(np.sort(w), np.linalg.eigvalsh(T), atol=1e-12)
