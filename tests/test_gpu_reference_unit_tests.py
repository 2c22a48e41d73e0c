"""The reference's kernel-level unit tests (tests/linalg/internal/cuda/*.cpp, the GPU twins of cpu/*.cpp) replayed on the HIP
path through the Impl's ChaseBase virtuals: same matrices, same calls, the reference's own assertions and tolerances.
(HEMM known answer, CholQR / Householder fixtures, flipSign, symOrHerm, absTrace / shiftDiagonal / lacpy known answers and the
pseudo-Hermitian Lanczos / Rayleigh-Ritz tests live in test_gpu_kernels.py, test_gpu_solve.py and test_gpu_pseudo.py.)"""
import numpy as np
import pytest
from oracle import chase_oracle as O

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def _rand_unitary(N, cplx):
    """Orthonormal basis from N(0,1) entries of mt19937(1337), like the reference's fixtures (geqrf + gqr of a random matrix;
    any orthonormal basis serves the assertions, which only use H = Q D Q^H and LAPACK's eigenpairs of it)."""
    g = O.StdNormal(1337)
    d = g.draw(2 * N * N if cplx else N * N)
    X = (d[0::2] + 1j * d[1::2] if cplx else d).reshape((N, N), order="F")
    Q, _ = np.linalg.qr(X)
    return Q


@pytest.mark.parametrize("cplx", [False, True])
def test_rayleigh_ritz_eigenpairs(ctx, cplx):
    """cuda/rayleighRitz.cpp:56-131 (cpu/rayleighRitz.cpp:48-118): H = Q diag(0.1 (i+1)) Q^H, N = 50, n = 10; V1 = the first n
    eigenvectors of H (heevd), rayleighRitz(H, V1, V2, ritzv, offset 2, subSize 5): the Ritz values of columns 2..6 equal
    LAPACK's eigenvalues within 100 eps."""
    from chase_amd.capi import Solver
    N, n, offset, sub = 50, 10, 2, 5
    Q = _rand_unitary(N, cplx)
    H = (Q * (0.1 * np.arange(N) + 0.1)[None, :]) @ Q.conj().T
    H = np.asfortranarray((H + H.conj().T) / 2)
    evals, evecs = np.linalg.eigh(H)
    V = np.asfortranarray(evecs[:, :n].astype(H.dtype))
    s = Solver(ctx, H, n - 2, 2, V=V)
    s.Start(); s.initVecs(False)                    # the caller's V is the start block (chase_gpu.hpp:527-541)
    s.Lock(offset)
    s.RR(sub, offset)
    assert np.max(np.abs(s.ritzv[offset:offset + sub] - evals[offset:offset + sub])) <= 100 * EPS
    s.close()


@pytest.mark.parametrize("cplx", [False, True])
def test_residuals_diagonal_and_dense(ctx, cplx):
    """cuda/residuals.cpp:57-96: diagonal H = diag(1..64), unit vectors, exact eigenvalues -> every residual within 10 eps of
    eps; :98-163: H = Q diag(0.1 (i+1)) Q^H with LAPACK's eigenpairs, residuals of columns 2..11 within 100 eps of eps."""
    from chase_amd.capi import Solver
    N = 64
    dt = np.complex128 if cplx else np.float64
    H = np.asfortranarray(np.diag(np.arange(1.0, N + 1)).astype(dt))
    V = np.asfortranarray(np.eye(N, dtype=dt))
    s = Solver(ctx, H, N - 1, 1, V=V)
    s.Start(); s.initVecs(False)
    s.ritzv[:] = np.arange(1.0, N + 1)
    r = s.Resd(0)
    assert np.all(np.abs(r - EPS) <= 10 * EPS)
    s.close()
    Q = _rand_unitary(N, cplx)
    H = (Q * (0.1 * np.arange(N) + 0.1)[None, :]) @ Q.conj().T
    H = np.asfortranarray((H + H.conj().T) / 2)
    evals, evecs = np.linalg.eigh(H)
    V = np.asfortranarray(evecs.astype(dt))
    s = Solver(ctx, H, N - 1, 1, V=V)
    s.Start(); s.initVecs(False)
    s.ritzv[:] = evals
    offset, sub = 2, 10
    s.Lock(offset)
    r = s.Resd(offset)                              # residuals of columns offset.. (the Impl uses locked_, chase_cpu.hpp:805-818)
    assert np.all(np.abs(r[:sub] - EPS) <= 100 * EPS)
    s.close()


@pytest.mark.parametrize("cplx", [False, True])
def test_lanczos_bounds_on_clement(ctx, cplx):
    """cuda/lanczos.cpp:70-116 (cpu/lanczos.cpp): Clement matrix N = 500 (spectrum -(N-1) .. N-1), M = 10 steps.  mlanczos
    with 4 vectors: every run's smallest Ritz value > 1 - N, largest < N - 1, and N - 1 < upperb < 5 (N - 1); the
    single-vector form: the same bound on upperb."""
    from chase_amd.capi import Solver
    N, M, numvec = 500, 10, 4
    dt = np.complex128 if cplx else np.float64
    H = np.zeros((N, N), dtype=dt, order="F")
    i = np.arange(N - 1)
    off = np.sqrt(i * (N + 1.0 - i))                # the reference fixture's own (0-based) entries, lanczos.cpp:38-46
    H[i + 1, i] = off; H[i, i + 1] = off
    s = Solver(ctx, H, 8, 4)
    s.Start(); s.initVecs(True)
    ub, theta, tau, ritzV = s.Lanczos(M, numvec)
    th = theta.reshape(numvec, M)
    assert np.all(th[:, 0] > 1.0 - N) and np.all(th[:, M - 1] < N - 1.0)
    assert N - 1 < ub < 5 * (N - 1)
    s.initVecs(True)
    ub1 = s.Lanczos(M, 0)
    assert N - 1 < ub1 < 5 * (N - 1)
    s.close()
