"""Pins the CPU oracle (oracle/chase_oracle.py) against the known-answer tests and fixtures the reference's own test
suite holds for the hot path.  CPU only."""
import math
import numpy as np
import pytest
from conftest import read_ref_matrix
from oracle import chase_oracle as O

EPS = np.finfo(np.float64).eps


def test_mt19937_known_answer():
    # ISO C++ [rand.predef]: the 10000th invocation of a default-constructed mt19937 produces 4123659995
    assert int(O.StdNormal(5489).raw(10000)[-1]) == 4123659995


def test_normal_stream_is_chunking_invariant():
    a = O.StdNormal(1337)
    x = np.concatenate([a.draw(7), a.draw(4), a.draw(1), a.draw(100)])
    y = O.StdNormal(1337).draw(112)
    assert np.array_equal(x, y)
    assert abs(y.mean()) < 0.3 and 0.7 < y.std() < 1.3


def test_hemm_known_answer_49_986():
    # tests/linalg/internal/mpi/hemm.cpp:36-119: H == 1 (10x10), V == 2, W == 3, alpha 2, beta 3, 2 of 4 columns
    k = O.OracleCPU(np.ones((10, 10), order="F"), 2, 2)
    k.V1[:] = 2.0
    k.V2[:] = 3.0
    k.HEMM(2, 2.0, 3.0, 0)          # V2 = 2*H*V1 + 3*V2 on two columns, then swap
    assert np.all(k.V1[:, :2] == 49.0) and np.all(k.V1[:, 2:] == 3.0)
    k.HEMM(2, 2.0, 3.0, 0)
    assert np.all(k.V1[:, :2] == 986.0) and np.all(k.V1[:, 2:] == 2.0)


@pytest.mark.parametrize("cplx", [False, True])
def test_cholqr_reference_fixtures(cplx):
    # tests/linalg/internal/cpu/cholqr1.cpp:31-126 (m = 100, n = 50)
    pre = "matrix_cdouble_" if cplx else "matrix_double_"
    V10 = read_ref_matrix(pre + "cond_10.bin", 100, 50, cplx)
    V1e4 = read_ref_matrix(pre + "cond_1e4.bin", 100, 50, cplx)
    Vill = read_ref_matrix(pre + "cond_ill.bin", 100, 50, cplx)
    Q, info = O.cholQR1(V10)
    assert info == 0 and abs(O.orthogonality(Q) - EPS) <= 15 * EPS
    Q, info = O.cholQR1(V1e4)
    assert info == 0 and EPS < O.orthogonality(Q) < 1.0
    _, info = O.cholQR1(Vill)
    assert 0 < info <= 50
    Q, info = O.cholQR2(V1e4)
    assert info == 0 and abs(O.orthogonality(Q) - EPS) <= 15 * EPS
    _, info = O.cholQR2(Vill)
    assert 0 < info <= 50
    Q, info = O.shiftedcholQR2(Vill)
    assert info == 0 and abs(O.orthogonality(Q) - EPS) <= 10 * EPS
    # Householder (tests/linalg/internal/mpi/householder_qr.cpp:46-93: 25 eps)
    assert O.orthogonality(O.houseQR(Vill)) <= 25 * EPS


@pytest.mark.parametrize("cplx", [False, True])
def test_rayleigh_ritz_known_spectrum(cplx):
    # tests/linalg/internal/cpu/rayleighRitz.cpp:48-118: H = Q diag(0.1 (i+1)) Q^H, N = 50, n = 10
    N, n = 50, 10
    X = O.random_start_vectors(N, N, cplx)
    Qf, _ = np.linalg.qr(X)
    lam = 0.1 * (np.arange(N) + 1)
    H = (Qf * lam[None, :]) @ Qf.conj().T
    H = np.asfortranarray((H + H.conj().T) / 2)
    w, V = O.rayleighRitz(H, Qf[:, :n])
    assert np.max(np.abs(w - lam[:n])) <= 100 * EPS
    assert np.max(O.residuals(H, w, V)) <= 1e3 * EPS


def test_residuals_diagonal_case():
    # tests/linalg/internal/cpu/residuals.cpp:40-80: diagonal H, unit vectors -> zero residuals
    N = 20
    H = np.diag(np.arange(1.0, N + 1))
    V = np.eye(N)[:, :5]
    assert np.max(O.residuals(H, np.arange(1.0, 6), V)) <= EPS


@pytest.mark.parametrize("cplx", [False, True])
def test_clement_solve_n256(cplx):
    # tests/chase_serial_solve.cpp:36-190: N = 256, nev = 24, nex = 16, tol 1e-10, deg 16 -> residuals < 1e-8
    H = O.clement(256, cplx)
    assert np.allclose(H, H.conj().T)
    k = O.OracleCPU(H, 24, 16)
    k.config.deg = 16
    st = O.solve(k)
    assert st["iterations"] < 25
    assert np.all(np.isfinite(k.ritzv[:24]))
    assert np.max(k.resid[:24]) < 1e-8
    assert np.max(O.residuals(H, k.ritzv[:24], k.V1[:, :24])) < 1e-8
    assert np.all(np.diff(k.ritzv[:24]) >= 0)
    # spectrum of this Clement variant: -N, -N+2, ... (measured with the reference)
    assert np.max(np.abs(k.ritzv[:5] - (-256 + 2 * np.arange(5)))) < 1e-4


def test_clement_solve_n1001():
    # tests/chase_distributed_solve.cpp:209-284 shape: N = 1001, nev = 100, nex = 60
    H = O.clement(1001, False)
    k = O.OracleCPU(H, 100, 60)
    st = O.solve(k)
    assert st["iterations"] < 25
    assert np.max(k.resid[:100]) < 1e-8
    assert np.max(O.residuals(H, k.ritzv[:100], k.V1[:, :100])) < 1e-8


def test_pseudo_hermitian_oracle_on_reference_bse_fixture():
    # tests/chase_serial_solve_pseudo_bse_test.cpp:104-224 + the fixture's reference spectrum
    import os
    from conftest import REF_FIX
    H = read_ref_matrix("cdouble_random_BSE.bin", 200, 200, True)
    eigs = np.fromfile(os.path.join(REF_FIX, "eigs_cdouble_random_BSE.bin"), dtype=np.complex128).real
    pos = np.sort(eigs[eigs > 0])
    k = O.OraclePseudoCPU(H, 20, 20)
    k.config.num_lanczos, k.config.lanczos_iter = 10, 50
    st = O.solve_pseudo(k)
    assert st["locked"] >= 20 and st["iterations"] < 25
    assert np.max(k.resid[:20]) <= 1e-10
    assert np.max(O.residuals(H, k.ritzv[:20], k.V1[:, :20])) <= 1e-10
    assert np.max(np.abs(k.ritzv[:20] - pos[:20])) <= 1e-10


@pytest.mark.parametrize("tag,N", [("cdouble_tiny_random_BSE", 10), ("cdouble_random_BSE", 200)])
def test_pseudo_lanczos_reference_assertions(tag, N):
    """tests/linalg/internal/cpu/pseudo_hermitian_lanczos.cpp:95-199 on the reference's own fixtures: full-length
    S-inner-product Lanczos (M = N, one vector, mt19937(1337) start block) reproduces the extreme eigenvalues to
    diff^2 < 1e3 eps, and the bound of the single-vector overload lies in [0.99, 5) x the largest eigenvalue."""
    import os
    from conftest import REF_FIX
    H = read_ref_matrix(tag + ".bin", N, N, True)
    eigs = np.fromfile(os.path.join(REF_FIX, "eigs_%s.bin" % tag), dtype=np.complex128).real
    k = O.OraclePseudoCPU(H, N // 2 - N // 4, N // 4)
    k.Start()
    k.V1 = O.random_start_vectors(N, N, True)
    ub, theta, tau, ritzV = k.Lanczos(N, 1)
    eps = np.finfo(np.float64).eps
    assert (theta[0] - eigs[0]) ** 2 < 1e3 * eps and (theta[N - 1] - eigs[N - 1]) ** 2 < 1e3 * eps
    assert ub == theta[N - 1]
    k.V1 = O.random_start_vectors(N, N, True)
    ub1 = k.Lanczos(N)
    assert (ub1 >= eigs[N - 1] or abs(ub1 - eigs[N - 1]) / abs(eigs[N - 1]) <= 1e-2) and ub1 < 5 * eigs[N - 1]


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


def test_lanczos_for_H2_reference_assertions():
    import os
    from conftest import REF_FIX
    n, nev, nex, numvec, m = 200, 20, 20, 10, 50
    H = read_ref_matrix("cdouble_random_BSE.bin", n, n, True)
    eigs = np.fromfile(os.path.join(REF_FIX, "eigs_cdouble_random_BSE.bin"), dtype=np.complex128).real
    k = O.OraclePseudoCPU(H, nev, nex)
    k.config.num_lanczos, k.config.lanczos_iter = numvec, m
    k.Start()
    k.V1 = _lanczos_for_H2_start_block(n, nev + nex, k.ncol)
    k.initVecs(False)
    k.QR(0, 1.0)
    upperb, idx = O.lanczos_for_H2(k, n, numvec, m, nev + nex, k.ritzv)
    _lanczos_for_H2_reference_assertions(eigs, nev + nex, m, upperb, idx, k.ritzv)


@pytest.mark.parametrize("N,cplx,nev,nex,iters,vecs", [(4096, False, 100, 40, 8, 24988), (1200, True, 80, 60, 5, 12664)])
def test_oracle_reproduces_the_survey_cross_check_counts(N, cplx, nev, nex, iters, vecs):
    """Survey cross-check counts (the survey's cross-check table, BASELINE.md: counts printed by a build of the reference that needed a
    hand-written fortran_mangle.h, which this repository may not use as a pin - they are consistency counts, the pins are the
    reference's fixtures and its own driver):
    ChASECPU<double> on the unperturbed Clement matrix N = 4096, nev = 100, nex = 40 -> 8 iterations, 24 988 filtered vectors;
    tests/noinput.cpp problem #0 (ChASECPU<complex<double>>, N = 1200, nev = 80, nex = 60) -> 5 iterations, 12 664 vectors;
    eigenvalues -N, -N+2, ..., residuals below 1e-10.  Defaults: tol 1e-10, deg 20, opt on, mt19937(1337) start block."""
    k = O.OracleCPU(O.clement(N, cplx, perturb=0), nev, nex)
    so = O.solve(k)
    assert (so["iterations"], so["filtered_vecs"]) == (iters, vecs)
    assert np.max(np.abs(k.ritzv[:nev] - (-N + 2.0 * np.arange(nev)))) < 1e-8
    assert np.max(k.resid[:nev]) <= 1e-10


def test_oracle_in_its_distributed_form_reproduces_the_survey_cross_check_counts():
    """Survey cross-check counts of examples/1_hello_world (the survey's cross-check table, BASELINE.md: counts printed by a build of the reference that needed a
    hand-written fortran_mangle.h, which this repository may not use as a pin - they are consistency counts, the pins are the
    reference's fixtures and its own driver): pChASECPU, unperturbed
    complex Clement N = 1200, nev = 80, nex = 60, block-cyclic nb = 64 on a 2 x 2 grid -> 6 iterations, 13 310 filtered vectors.
    The oracle follows pChASECPU where the two reference Impls differ for the driver (start vectors from mt19937(1337 + grid
    row) per block of local rows, V2 refreshed by QR, Swap on both blocks); in its ChASECPU form the same problem takes 5
    iterations / 12 664 vectors (test above), like the reference's sequential binary."""
    from chase_amd import dist as cd
    N, nev, nex, nb = 1200, 80, 60, 64
    rl = cd.Layout(N, nb, 2)
    k = O.OracleCPU(O.clement(N, True, perturb=0), nev, nex, grid_rows=[rl.globals_of(i) for i in range(2)])
    so = O.solve(k)
    assert (so["iterations"], so["filtered_vecs"]) == (6, 13310)
    assert np.max(np.abs(k.ritzv[:nev] - (-N + 2.0 * np.arange(nev)))) < 1e-8


@pytest.mark.parametrize("tag,N", [("cdouble_tiny_random_BSE", 10), ("cdouble_random_BSE", 200)])
def test_flip_lower_half_gives_the_fixture_SH_spectrum(tag, N):
    """S H (lower half of the rows negated, flipLowerHalfMatrixSign) is Hermitian positive definite with the spectrum the
    reference stores next to its BSE fixtures (SH_eigs_*.bin)."""
    import os
    from conftest import REF_FIX
    H = read_ref_matrix(tag + ".bin", N, N, True)
    want = np.sort(np.fromfile(os.path.join(REF_FIX, "SH_eigs_%s.bin" % tag), dtype=np.float64))      # N real values
    SH = O.flip_lower_half(H.copy())
    assert np.linalg.norm(SH - SH.conj().T) <= 1e-12 * np.linalg.norm(SH)
    got = np.linalg.eigvalsh((SH + SH.conj().T) / 2)
    assert got.min() > 0 and np.max(np.abs(got - want)) <= 1e-10 * np.abs(want).max()
