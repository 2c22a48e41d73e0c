"""The library's own small eigensolvers (chase_amd/csrc/host_eig_builtin.cpp), the provider of last resort when no host
LAPACK can be bound: selected here with CHASE_HIP_LAPACK_LIB=builtin in a child process (the provider is bound once per
process), checked against numpy with the reference's Rayleigh-Ritz tolerances (tests/linalg/internal/cpu/rayleighRitz.cpp:
eigenvalues to 100 eps relative to the spectrum, orthogonality / residual of the eigenvectors)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import ctypes as C, numpy as np, sys
    sys.path.insert(0, %r)
    from chase_amd.capi import lib
    assert lib.chase_hip_lapack_provider() == b"builtin", lib.chase_hip_lapack_provider()
    eps = np.finfo(float).eps
    rng = np.random.default_rng(5)
    for cplx in (0, 1):
        for n in (1, 2, 3, 17, 96, 200):
            A = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
            A = np.asfortranarray((A + A.conj().T) / 2)
            if n >= 17:                         # a cluster and an exactly repeated eigenvalue
                Q, _ = np.linalg.qr(A)
                lam = np.linspace(-3, 5, n); lam[3] = lam[4]; lam[7] = lam[8] * (1 + 1e-13)
                A = np.asfortranarray((Q * lam) @ Q.conj().T); A = np.asfortranarray((A + A.conj().T) / 2)
            want = np.linalg.eigvalsh(A)
            Z = A.copy(order="F"); w = np.zeros(n)
            # garbage in the strictly upper triangle: only the lower one may be referenced
            Z[np.triu_indices(n, 1)] = 7.0
            rc = lib.chase_hip_heevd_host(cplx, n, Z.ctypes.data, n, w.ctypes.data)
            assert rc == 0, (rc, lib.chase_hip_last_error())
            scale = max(1.0, np.abs(want).max())
            assert np.all(np.diff(w) >= 0) and np.max(np.abs(w - want)) <= 100 * eps * scale, (cplx, n, np.max(np.abs(w - want)))
            assert np.linalg.norm(Z.conj().T @ Z - np.eye(n)) <= 50 * eps * n
            assert np.linalg.norm(A @ Z - Z * w) <= 100 * eps * n * scale
    for n in (1, 2, 25, 300, 700):
        d, e = rng.standard_normal(n), np.abs(rng.standard_normal(n))
        T = np.diag(d) + np.diag(e[:n - 1], 1) + np.diag(e[:n - 1], -1)
        want = np.linalg.eigvalsh(T)
        dd, ee, w, Z = d.copy(), e.copy(), np.zeros(n), np.zeros((n, n), order="F")
        rc = lib.chase_hip_stemr_host(n, dd.ctypes.data, ee.ctypes.data, w.ctypes.data, Z.ctypes.data, n)
        assert rc == 0
        assert np.max(np.abs(w - want)) <= 100 * eps * max(1.0, np.abs(want).max())
        assert np.linalg.norm(Z.T @ Z - np.eye(n)) <= 50 * eps * n
        assert np.linalg.norm(T @ Z - Z * w) <= 100 * eps * n * max(1.0, np.abs(want).max())
    print("BUILTIN_OK")
''') % ROOT


def test_builtin_eigensolvers_match_numpy():
    env = dict(os.environ, CHASE_HIP_LAPACK_LIB="builtin")
    p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "BUILTIN_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
