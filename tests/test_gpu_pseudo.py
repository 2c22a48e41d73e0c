"""Pseudo-Hermitian (Bethe-Salpeter) path — SURVEY.md §8 row A11 / BASELINE config 5: HEMM_H2, K-conjugation,
S-orthogonal QR, rayleighRitz_v2 and chase::Solve_pseudo of the HIP Impl against the CPU oracle and against the reference's
own BSE fixture (tests/chase_serial_solve_pseudo_bse_test.cpp:104-224).  GPU only."""
import numpy as np
import pytest
from conftest import read_ref_matrix
from oracle import chase_oracle as O

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def bse_fixture():
    H = read_ref_matrix("cdouble_random_BSE.bin", 200, 200, True)
    eigs = np.fromfile(__import__("os").path.join(__import__("conftest").REF_FIX, "eigs_cdouble_random_BSE.bin"),
                       dtype=np.complex128).real
    return H, np.sort(eigs[eigs > 0])


def test_bse_fixture_is_pseudo_hermitian():
    H, pos = bse_fixture()
    S = np.diag(np.r_[np.ones(100), -np.ones(100)])
    SH = S @ H
    assert np.linalg.norm(SH - SH.conj().T) <= 1e-12 * np.linalg.norm(SH)
    assert np.all(np.linalg.eigvalsh((SH + SH.conj().T) / 2) > 0)
    assert len(pos) == 100


def test_pseudo_operators_vs_oracle(ctx):
    from chase_amd.capi import PseudoSolver
    H, _ = bse_fixture()
    nev, nex = 12, 8
    ne = nev + nex
    s = PseudoSolver(ctx, H, nev, nex)
    k = O.OraclePseudoCPU(H, nev, nex)
    s.Start(); k.Start()
    s.initVecs(True); k.initVecs(True)
    assert np.array_equal(s.peek_v(), k.V1)
    s.QR(0, 1.0); k.QR(0, 1.0)
    assert np.max(np.abs(s.peek_v() - k.V1)) < 1e-12
    # H^2 filter steps incl. the gamma term and a column offset
    for (a, b, g, off) in [(1e-3, 0.0, -0.2, 0), (2e-3, -0.3, -0.4, 0), (2e-3, -0.25, -0.4, 3)]:
        s.HEMM_H2(ne, a, b, g, off); k.HEMM_H2(ne, a, b, g, off)
    Vg, Vo = s.peek_v(), k.V1
    assert np.max(np.abs(Vg[:, 3:ne] - Vo[:, 3:ne])) <= 1e-11 * np.abs(Vo).max()
    s.HEMM_H2(0, 0, 0, 0, 0); k.HEMM_H2(0, 0, 0, 0, 0)           # even number of swaps
    s.ApplyKconjugate(ne); k.ApplyKconjugate(ne)
    Vg, Vo = s.peek_v(), k.V1
    assert np.max(np.abs(Vg[:, ne:] - Vo[:, ne:])) <= 1e-11 * np.abs(Vo).max()
    # K-conjugate structure: second half = [conj(lower); conj(upper)] of the first half
    assert np.array_equal(Vg[100:, ne:], np.conj(Vg[:100, :ne])) and np.array_equal(Vg[:100, ne:], np.conj(Vg[100:, :ne]))
    # restart from a well-conditioned K-symmetric block for QR / RR / Resd
    s.initVecs(True); k.initVecs(True)
    s.QR(0, 1.0); k.QR(0, 1.0)
    s.ApplyKconjugate(ne); k.ApplyKconjugate(ne)
    s.QR(0, 1e3); k.QR(0, 1e3)
    assert s.get("qr_variant") == k.qr_variant
    assert np.max(np.abs(s.peek_v() - k.V1)) < 1e-10
    s.RR(ne, 0); k.RR(k.ritzv, ne)
    assert np.max(np.abs(s.ritzv - k.ritzv)) <= 1e-9 * np.abs(k.ritzv).max()
    r_g = s.Resd(0)
    r_o = np.zeros(ne); k.Resd(k.ritzv, r_o, 0)
    assert np.max(np.abs(r_g - r_o)) <= 1e-9 * max(1.0, r_o.max())
    s.close()


def test_solve_pseudo_bse_fixture(ctx):
    """The reference's integration test: n = 200, nev = nex = 20, numLanczos = 10, lanczosIter = 50, tol 1e-10."""
    from chase_amd.capi import PseudoSolver
    H, pos = bse_fixture()
    nev, nex = 20, 20
    s = PseudoSolver(ctx, H, nev, nex)
    s.set(tol=1e-10, deg=20, opt=1, maxiter=25, numlanczos=10, lanczositer=50)
    st = s.solve(trace=True)
    lam = s.ritzv[:nev].copy()
    resid = s.resid()[:nev]
    assert np.all(np.isfinite(lam)) and np.all(np.isfinite(resid))
    assert np.max(resid) <= 1e-10                                         # the reference's assertion
    V = s.V[:, :nev]
    r_host = np.linalg.norm(H @ V - V * lam[None, :], axis=0)
    assert np.max(r_host) <= 1e-10                                         # recomputed like the reference test
    assert np.max(np.abs(s.recompute_residuals(nev) - r_host)) <= 1e-12    # the library's own independent check
    assert np.max(np.abs(lam - pos[:nev])) <= 1e-9                         # fixture spectrum: smallest positive eigenvalues
    k = O.OraclePseudoCPU(H, nev, nex); k.config.num_lanczos = 10; k.config.lanczos_iter = 50
    so = O.solve_pseudo(k)
    assert np.max(np.abs(lam - k.ritzv[:nev])) <= 1e-9
    assert abs(st["iterations"] - so["iterations"]) <= 1
    assert st["locked"] >= nev
    s.close()


def test_gen_bse_structure_and_shards(ctx):
    """chase_hip_gen_bse: H = [[A, B], [-conj(B), -conj(A)]], S H Hermitian positive definite; block-cyclic shards
    generated independently equal the corresponding entries of the whole matrix."""
    from chase_amd import dist as cd
    N, h = 96, 48
    H = ctx.gen_bse(N, True, dmin=1.0, dmax=5.0, offdiag=1e-2, seed=11).download()
    A, B = H[:h, :h], H[:h, h:]
    assert np.array_equal(A, A.conj().T) and np.array_equal(B, B.T)
    assert np.array_equal(H[h:, :h], -B.conj()) and np.array_equal(H[h:, h:], -A.conj())
    assert np.allclose(np.diag(A).real, np.sqrt(1.0 + 24.0 * np.arange(h) / (h - 1)), rtol=0, atol=1e-14)
    SH = np.vstack([H[:h], -H[h:]])
    assert np.linalg.norm(SH - SH.conj().T) == 0 and np.linalg.eigvalsh(SH).min() > 0.5
    off = A[np.triu_indices(h, 1)]
    assert 0.5e-2 < off.real.std() < 2e-2 and 0.5e-2 < off.imag.std() < 2e-2
    for (mb, pr, pc) in [(0, 2, 2), (8, 3, 2)]:
        rl, cl = cd.Layout(N, mb, pr), cd.Layout(N, mb, pc)
        for i in range(pr):
            for j in range(pc):
                blk = cd.gen_bse_local(ctx, N, True, rl, cl, i, j, dmin=1.0, dmax=5.0, offdiag=1e-2, seed=11).download()
                assert np.array_equal(blk, H[np.ix_(rl.globals_of(i), cl.globals_of(j))])
    Hr = ctx.gen_bse(N, False, dmin=1.0, dmax=5.0, offdiag=1e-2, seed=11).download()
    assert np.array_equal(Hr[:h, :h], Hr[:h, :h].T) and np.array_equal(Hr[h:, h:], -Hr[:h, :h])
    assert np.array_equal(Hr[h:, :h], -Hr[:h, h:])


def test_solve_pseudo_generated_bse_vs_oracle(ctx):
    """Solve_pseudo on the synthetic BSE matrix of the benchmark family (N = 400) against the oracle and numpy."""
    from chase_amd.capi import PseudoSolver
    N, nev, nex = 400, 16, 10
    H = np.asfortranarray(ctx.gen_bse(N, True, dmin=1.0, dmax=11.0, offdiag=1e-3, seed=7).download())
    s = PseudoSolver(ctx, H, nev, nex)
    s.set(tol=1e-10, numlanczos=10, lanczositer=50)
    st = s.solve()
    lam = s.ritzv[:nev].copy()
    ev = np.linalg.eigvals(H).real
    pos = np.sort(ev[ev > 0])
    assert np.max(np.abs(np.sort(lam) - pos[:nev])) <= 1e-9
    assert np.max(s.resid()[:nev]) <= 1e-10
    k = O.OraclePseudoCPU(H, nev, nex); k.config.num_lanczos = 10; k.config.lanczos_iter = 50
    so = O.solve_pseudo(k)
    assert np.max(np.abs(np.sort(lam) - np.sort(k.ritzv[:nev]))) <= 1e-9
    assert abs(st["iterations"] - so["iterations"]) <= 1
    s.close()


@pytest.mark.parametrize("tag,N", [("cdouble_tiny_random_BSE", 10), ("cdouble_random_BSE", 200)])
def test_pseudo_lanczos_reference_assertions_on_gpu(ctx, tag, N):
    """The reference's pseudo-Hermitian Lanczos tests (tests/linalg/internal/cpu/pseudo_hermitian_lanczos.cpp:95-199,
    cuda/pseudo_hermitian_lanczos.cpp) through the HIP Impl's Lanczos virtuals, same fixtures and assertions."""
    import os
    from conftest import REF_FIX
    from chase_amd.capi import PseudoSolver
    from oracle import chase_oracle as O
    H = read_ref_matrix(tag + ".bin", N, N, True)
    eigs = np.fromfile(os.path.join(REF_FIX, "eigs_%s.bin" % tag), dtype=np.complex128).real
    s = PseudoSolver(ctx, H, N // 2 - N // 4, N // 4)        # 2 (nev + nex) = N columns hold the M = N Krylov vectors
    s.Start()
    s.V[:] = O.random_start_vectors(N, N, True)
    s.initVecs(False)
    ub, theta, tau, ritzV = s.Lanczos(N, 1)
    eps = np.finfo(np.float64).eps
    assert (theta[0] - eigs[0]) ** 2 < 1e3 * eps and (theta[N - 1] - eigs[N - 1]) ** 2 < 1e3 * eps
    assert ub == theta[N - 1]
    s.V[:] = O.random_start_vectors(N, N, True)
    s.initVecs(False)
    ub1 = s.Lanczos(N, 0)
    assert (ub1 >= eigs[N - 1] or abs(ub1 - eigs[N - 1]) / abs(eigs[N - 1]) <= 1e-2) and ub1 < 5 * eigs[N - 1]
    s.close()


def _lanczos_for_H2_reference_assertions(eigs, nevex, m, upperb, idx, ritzv):
    """tests/algorithm/lanczos_for_H2_test.cpp:103-232: properties of the H^2 bounds on the BSE fixture."""
    n = len(eigs)
    e2 = np.sort(eigs ** 2)
    smallest, largest = e2[0], e2[n - 1]
    lam_2nevex = e2[2 * nevex - 1] if 2 * nevex - 1 < n else e2[n - 1]
    assert np.isfinite(upperb) and upperb > 0
    assert np.all(np.isfinite(ritzv[:nevex])) and np.all(ritzv[:nevex] >= 0)
    mu_1 = min(ritzv[: nevex - 1])
    mu_nn = ritzv[nevex - 1]
    assert 0 < mu_1 < mu_nn
    gap_low = lam_2nevex - smallest
    if gap_low > 0:
        assert abs(mu_1 - smallest) <= 0.2 * gap_low
    assert mu_nn > 0 and mu_nn >= lam_2nevex and mu_nn <= upperb
    if abs(mu_nn - largest) > 0:
        assert abs(mu_nn - lam_2nevex) <= 0.35 * abs(mu_nn - largest)
    assert 0.98 * largest <= upperb <= 1.02 * largest
    assert idx <= m


def _lanczos_for_H2_start_block(n, nevex, ncol):
    """per-column generators mt19937(1314521 + j), T(dist(gen), dist(gen)) like the reference test"""
    V = np.zeros((n, ncol), dtype=np.complex128, order="F")
    for j in range(nevex):
        d = O.StdNormal(1314521 + j).draw(2 * n)
        V[:, j] = d[0::2] + 1j * d[1::2]
    return V


def test_lanczos_for_H2_reference_assertions_on_gpu(ctx):
    """The reference's algorithm test (LanczosForH2_GPU_ReturnsH2Bounds) through the HIP Impl and the own driver."""
    import os
    from conftest import REF_FIX
    from chase_amd.capi import PseudoSolver
    n, nev, nex, numvec, m = 200, 20, 20, 10, 50
    H = read_ref_matrix("cdouble_random_BSE.bin", n, n, True)
    eigs = np.fromfile(os.path.join(REF_FIX, "eigs_cdouble_random_BSE.bin"), dtype=np.complex128).real
    s = PseudoSolver(ctx, H, nev, nex)
    s.set(numlanczos=numvec, lanczositer=m)
    s.Start()
    s.V[:] = _lanczos_for_H2_start_block(n, nev + nex, s.ncol)
    s.initVecs(False)
    s.QR(0, 1.0)
    upperb, idx = s.lanczos_for_H2(numvec, m)
    _lanczos_for_H2_reference_assertions(eigs, nev + nex, m, upperb, idx, s.ritzv)
    s.close()


def test_real_pseudo_hermitian_fixture_solve(ctx):
    """The `double` instantiation of the pseudo-Hermitian path (ChASEGPU<double, PseudoHermitianMatrix>): the reference's
    real BSE fixture [[A, B], [-B, -A]] (200 x 200) solved with ChaseHipPseudo<double>; smallest positive eigenvalues of
    eigs_double_random_BSE.bin, residuals <= 1e-10, and the same iteration count as the oracle."""
    import os
    from conftest import REF_FIX
    from chase_amd.capi import PseudoSolver
    N, nev, nex = 200, 20, 20
    H = read_ref_matrix("double_random_BSE.bin", N, N, False)
    k = N // 2
    assert np.array_equal(H[k:, :k], -H[:k, k:]) and np.array_equal(H[k:, k:], -H[:k, :k])
    eigs = np.fromfile(os.path.join(REF_FIX, "eigs_double_random_BSE.bin"), dtype=np.float64)
    pos = np.sort(eigs[eigs > 0])
    s = PseudoSolver(ctx, H, nev, nex)
    assert not s.cplx
    s.set(tol=1e-10, numlanczos=10, lanczositer=50)
    st = s.solve()
    lam = s.ritzv[:nev].copy()
    assert np.max(np.abs(np.sort(lam) - pos[:nev])) <= 1e-9
    assert np.max(s.resid()[:nev]) <= 1e-10
    V = s.V[:, :nev]
    assert np.max(np.linalg.norm(H @ V - V * lam[None, :], axis=0)) <= 1e-9
    ko = O.OraclePseudoCPU(H.astype(np.complex128), nev, nex); ko.config.num_lanczos = 10; ko.config.lanczos_iter = 50
    so = O.solve_pseudo(ko)
    assert np.max(np.abs(np.sort(ko.ritzv[:nev]) - pos[:nev])) <= 1e-9
    assert abs(st["iterations"] - so["iterations"]) <= 2
    s.close()


@pytest.mark.parametrize("cplx,n", [(True, 640), (False, 400), (True, 96)])
def test_pseudo_rayleigh_ritz_dense_core(ctx, cplx, n):
    """chase_hip_pseudo_rr_small: the dense core of rayleighRitz_v2 (cpu/rayleighRitz.hpp:316-383: potrf, L^{-1} M L^{-H}, heevd,
    back-substitution, 1 / -w, normalisation).  For cores of 384 and more every step runs on the device (potrf_upper, one
    right-solve for R^{-1}, MFMA GEMMs, heevd_gpu - the reference's GPU path cuda/rayleighRitz.hpp:511-600), below that on the
    host LAPACK: both against scipy's generalised eigensolver.  n = 640 is config 5's core (2 (nev + nex))."""
    import scipy.linalg as sla
    from chase_amd.capi import lib, check
    rng = np.random.default_rng(17)
    def herm(scale):
        X = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
        return scale * (X + X.conj().T) / 2
    A = herm(0.05) + np.diag(np.linspace(1.0, 9.0, n))                     # Q^H S H Q: Hermitian positive definite
    M = herm(0.05) + np.diag(np.where(np.arange(n) % 2 == 0, 1.0, -1.0))  # Q^H S Q: Hermitian indefinite
    A = np.asfortranarray(A.astype(np.complex128 if cplx else np.float64))
    M = np.asfortranarray(M.astype(A.dtype))
    dA, dM = ctx.array(A), ctx.array(M)
    ritz = np.zeros(n)
    check(lib.chase_hip_pseudo_rr_small(ctx.h, int(cplx), n, dA.ptr, dM.ptr, ritz.ctypes.data), "pseudo_rr_small")
    X = dM.download()
    mu = sla.eigh(M, A, eigvals_only=True)                                # M x = mu A x, ascending
    want = 1.0 / mu[::-1]                                                  # w = -mu ascending  ->  ritz = 1 / -w
    assert np.max(np.abs(ritz - want) / np.abs(want)) < 1e-10
    h = n // 2
    assert np.max(np.abs(np.linalg.norm(X[:, :h], axis=0) - 1)) < 1e-12   # first n/2 vectors normalised
    R = A @ X[:, :h] - (M @ X[:, :h]) * ritz[:h]                           # A x = ritz M x
    assert np.max(np.linalg.norm(R, axis=0)) < 1e-10 * np.linalg.norm(A, 2) * np.max(np.abs(ritz[:h]))
    # not positive definite -> the potrf info comes back as a positive status
    A[5, 5] = -1.0
    dA.upload(A); dM.upload(M)
    rc = lib.chase_hip_pseudo_rr_small(ctx.h, int(cplx), n, dA.ptr, dM.ptr, ritz.ctypes.data)
    assert 0 < rc <= n
    dA.free(); dM.free()
