"""Distributed Impl (pChaseHip + grid collectives) on ONE GPU, in ONE process: the ranks of a grid are threads of the
pytest process (tests/rank_threads.py), each with its own context, stream and host-transport grid, so that every line of
the distributed C++ path except ncclAllReduce itself runs here on any grid shape - 2x1, 2x2, 3x2 and the 4x2 of BASELINE
configs[3] - while one process holds the GPU (the reference runs its distributed tests as ranks sharing one box too,
tests/CMakeLists.txt:23-31).  The RCCL transport is exercised on a 1x1 grid through size-1 communicators
(CHASE_HIP_RCCL_FORCE) in this process as well; real inter-device RCCL runs in bench.py --gpus N on the multi-GPU node.
Ranks as processes remain only at the end of the suite (tests/test_gpu_zz_processes.py)."""
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dist_scenarios as S  # noqa: E402
from rank_threads import run_ranks  # noqa: E402

pytestmark = pytest.mark.gpu
GRIDS = {1: (1, 1), 2: (2, 1), 4: (2, 2), 6: (3, 2), 8: (4, 2)}          # nprow >= npcol (grid/mpiGrid2D.hpp:209)


def run(nranks, fn, *args):
    run_ranks(*GRIDS[nranks], fn, *args)


def run_shared(nranks, fn, *args):
    """the same ranks-as-threads on the SHARED-DEVICE transport: device-side sums / copies between the ranks' buffers ordered by
    events between their streams (chase_hip_grid_create_shared) - asynchronous like RCCL, no host staging"""
    run_ranks(*GRIDS[nranks], fn, *args, transport="shared")


@pytest.mark.parametrize("nranks", [2, 4])
def test_hemm_known_answer(nranks):
    run(nranks, S.scenario_hemm_kat)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "d", 0), (4, "z", 0), (4, "z", 16), (2, "d", 32), (6, "z", 0), (8, "z", 16)])
def test_operators_vs_oracle(nranks, typ, mb):
    run(nranks, S.scenario_ops, typ == "z", mb)


def _panel_of_my_rank(ctx, grid, comm, N, nev, nex, out):
    import numpy as np
    from chase_amd import dist as cd
    rl, cl = cd.Layout(N, 0, grid.nprow), cd.Layout(N, 0, grid.npcol)
    m, n = rl.count(grid.myrow), cl.count(grid.mycol)
    dH = ctx.empty((m, n), np.complex128)                      # never read: only the object is built
    s = cd.DistSolver(ctx, grid, dH, N, nev, nex, True, 0, 0)
    out[comm.rank] = (m, n, int(s.get("panel_cols")))
    comm.barrier()
    s.close()
    dH.free()


def test_every_rank_derives_the_same_panel_grid():
    """The panels are what the pipelined all-reduces carry: every rank must arrive at the same width.  N = 32513 on two grid
    rows gives local blocks of 16257 and 16256 rows - 128 and 127 row tiles, for which "a panel fills the chip once" means 256
    and 320 -> 512 columns when taken from the rank's OWN block (rounds 2-4 did: a latent mismatch for such shapes); the width
    comes from the layout's largest block now."""
    out = {}
    run_ranks(2, 1, _panel_of_my_rank, 32513, 1200, 400, out)
    assert out[0][0] == 16257 and out[1][0] == 16256
    assert out[0][2] == out[1][2] == 256, out


@pytest.mark.parametrize("nranks,cplx,mb", [(4, True, 16), (8, False, 32)])
def test_pipeline_knobs_switched_between_iterations(nranks, cplx, mb):
    run(nranks, S.scenario_knob_switching, 640, 40, 24, cplx, mb, 20)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "d", 0), (4, "z", 0), (4, "z", 16), (8, "z", 0), (8, "d", 8), (6, "z", 0), (6, "d", 16),
                                           (2, "z", 32), (1, "d", 0)])
def test_distributed_sym_or_herm_matrix(nranks, typ, mb):
    run(nranks, S.scenario_sym_or_herm, typ == "z", mb)


def test_solve_block_2x2_n256_complex():
    run(4, S.scenario_solve, 256, 24, 16, True, 0, 16)


def test_solve_blockcyclic_2x2_n1001_nb64():
    # the reference's distributed integration test: N = 1001, nev = 100, nex = 60, nb = 64 on a 2 x 2 grid
    run(4, S.scenario_solve, 1001, 100, 60, False, 64, 20)


# ---- distributed pseudo-Hermitian (BSE) Impl: BASELINE config 5 / SURVEY.md §8 A11 ------------------------------------
@pytest.mark.parametrize("nranks,mb", [(2, 0), (4, 0), (4, 16), (6, 0), (8, 0)])
def test_pseudo_operators_vs_oracle(nranks, mb):
    run(nranks, S.scenario_pseudo_ops, mb)


@pytest.mark.parametrize("nranks,mb", [(4, 0), (4, 32), (2, 0)])
def test_pseudo_solve_bse_fixture(nranks, mb):
    run(nranks, S.scenario_pseudo_solve, mb)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "z", 0), (6, "d", 16), (1, "z", 0)])
def test_distributed_symmetry_check(nranks, typ, mb):
    run(nranks, S.scenario_symcheck, typ == "z", mb)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "z", 0), (2, "d", 0), (6, "z", 0), (8, "d", 0), (4, "z", 8), (6, "d", 16), (8, "z", 4),
                                           (1, "z", 0)])
def test_distributed_qr_on_reference_fixtures(nranks, typ, mb):
    """CholQR variants and the DISTRIBUTED Householder (panel factorisation over the row-distributed block: pivots cross
    rank boundaries with 6 and 8 ranks, block-cyclic rows with mb > 0) on the reference's conditioned fixtures"""
    run(nranks, S.scenario_qr_fixtures, typ == "z", mb)


def test_solve_blockcyclic_4x2_eight_ranks():
    """the 8-GPU grid shape of BASELINE configs[3] (4 x 2, block-cyclic nb = 64), eight ranks sharing one GPU"""
    run(8, S.scenario_solve, 1024, 100, 60, True, 64, 20)


def test_solve_block_3x2_six_ranks():
    run(6, S.scenario_solve, 700, 60, 40, False, 0, 20)


def test_pseudo_solve_4x2_eight_ranks():
    run(8, S.scenario_pseudo_solve, 0)


def test_distributed_run_reproduces_the_survey_cross_check_counts():
    """examples/1_hello_world in the survey's cross-check table (pChASECPU, 2 x 2, block-cyclic nb = 64): 6 iterations,
    13 310 filtered vectors - consistency counts, not pins (tests/test_oracle_pins.py)"""
    run(4, S.scenario_reference_run_counts)


@pytest.mark.parametrize("nranks,mb", [(4, 0), (2, 16)])
def test_pseudo_solve_real_fixture(nranks, mb):
    run(nranks, S.scenario_pseudo_solve_real, mb)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "d", 0), (4, "z", 0), (4, "z", 8), (2, "d", 16)])
def test_reference_distributed_kernel_tests(nranks, typ, mb):
    """tests/linalg/internal/mpi/{rayleighRitz,residuals,lanczos}.cpp (2 x 2 grid in the reference) through the grid Impl"""
    run(nranks, S.scenario_reference_units, typ == "z", mb)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "d", 0), (4, "z", 32), (2, "z", 0), (8, "z", 16)])
def test_distributed_c_entry_points(nranks, typ, mb):
    """p?chase_init[_blockcyclic]_hip_ / p?chase_ / p?chase_get_eigenpairs_ / p?chase_wrtHam_ / readHam_ / finalize_ (the
    interface's solver object belongs to the calling rank = thread); H filled after init and changed between solves"""
    run(nranks, S.scenario_cshim, typ == "z", mb)


@pytest.mark.parametrize("nranks", [4, 6, 8])
def test_grid_sendrecv_and_exact_agree_max(nranks):
    run(nranks, S.scenario_p2p)


@pytest.mark.parametrize("nprow,npcol,N,nev,nex,cplx,mb", [
    (3, 1, 301, 20, 10, False, 7),        # odd grid, block size that divides nothing
    (3, 2, 200, 60, 40, True, 0),         # half of the spectrum wanted, block layout with a short last block
    (2, 2, 130, 12, 1, True, 1),          # one extra vector, block-cyclic with 1 x 1 blocks
    (4, 1, 257, 30, 20, False, 64),       # last grid rows own a single partial block
    (2, 1, 96, 40, 40, False, 0),         # search space = 5/6 of the matrix
    (4, 1, 9, 2, 2, False, 0),            # the reference's block rule leaves the LAST grid row without rows: 3, 3, 3, 0
    (4, 2, 9, 2, 2, True, 0),             # ... and with two grid columns (5 + 4 columns)
    (5, 1, 16, 3, 3, True, 0),            # five grid rows: 4, 4, 4, 4, 0
])
def test_solve_on_awkward_grids_and_sizes(nprow, npcol, N, nev, nex, cplx, mb):
    from rank_threads import run_ranks as run_grid
    run_grid(nprow, npcol, S.scenario_solve, N, nev, nex, cplx, mb, 20, same_iterations=False)


@pytest.mark.parametrize("nprow,npcol,mb", [(3, 1, 0), (3, 2, 0), (4, 1, 0), (3, 1, 16), (2, 2, 1), (4, 2, 8)])
def test_pseudo_solve_on_awkward_grids(nprow, npcol, mb):
    """the K-conjugation partner rows (g + N/2) mod N land on other ranks in patterns the 2 x 2 / 4 x 2 cases do not produce"""
    from rank_threads import run_ranks as run_grid
    run_grid(nprow, npcol, S.scenario_pseudo_solve, mb)
    run_grid(nprow, npcol, S.scenario_pseudo_ops, mb)


@pytest.mark.parametrize("nprow,npcol,N,nev,nex,cplx,mb", [
    (2, 2, 256, 24, 16, True, 0), (2, 2, 1001, 100, 60, False, 64), (3, 1, 301, 20, 10, False, 7), (3, 2, 200, 60, 40, True, 0),
    (4, 1, 257, 30, 20, False, 64), (4, 2, 600, 40, 24, True, 16), (2, 1, 301, 20, 10, False, 0), (1, 1, 301, 20, 10, False, 0),
])
def test_grid_solver_takes_the_reference_distributed_path_count_for_count(nprow, npcol, N, nev, nex, cplx, mb):
    """iterations and filtered vectors EQUAL to the oracle following pChASECPU (start vectors per grid row, V2 refreshed by QR):
    the two reference Impls differ there, and on small problems the counts differ with them (3 x 1, N = 301: 10 iterations
    against the sequential Impl's 4) - the grid Impl must follow the distributed one"""
    from rank_threads import run_ranks as run_grid
    run_grid(nprow, npcol, S.scenario_solve_counts, N, nev, nex, cplx, mb, 20)


@pytest.mark.parametrize("nprow,npcol,mb", [(2, 2, 0), (4, 2, 0), (3, 1, 0), (2, 1, 16), (1, 1, 0)])
def test_pseudo_grid_solver_takes_the_reference_distributed_path_count_for_count(nprow, npcol, mb):
    from rank_threads import run_ranks as run_grid
    run_grid(nprow, npcol, S.scenario_pseudo_solve_counts, mb)


def test_c_interface_slot_is_process_wide_like_the_reference(ctx):
    """The reference keeps one static distributed solver per type for the PROCESS (chase_c_interface.cpp:905-1290): an
    application may call p?chase_init_ on one thread and p?chase_ / get_eigenpairs / finalize on another.  (Threads get slots of
    their own only after chase_hip_cshim_thread_ranks(1) - rank threads of one process: scenario_cshim.)"""
    import ctypes as C
    import threading
    import numpy as np
    from chase_amd.capi import lib
    from chase_amd import dist as cd
    from oracle import chase_oracle as O
    N, nev, nex = 200, 16, 12
    H = np.asfortranarray(O.clement(N, False))
    V = np.zeros((N, nev + nex), order="F")
    ritzv = np.zeros(nev + nex)
    grid = cd.Grid(ctx, 1, 1, 0, transport="host", pg=None)
    I = lambda v: C.byref(C.c_int(v))
    init = C.c_int(0)

    def on_worker():
        lib.chase_hip_cshim_use_ctx(C.c_void_p(ctx.h.value), 0)
        lib.pdchase_init_hip_(I(N), I(nev), I(nex), I(N), I(N), C.c_void_p(H.ctypes.data), I(N), C.c_void_p(V.ctypes.data),
                              C.c_void_p(ritzv.ctypes.data), C.c_void_p(grid.h.value), C.byref(init))

    t = threading.Thread(target=on_worker)
    t.start(); t.join()
    assert init.value == 1, lib.chase_hip_last_error()
    lib.chase_hip_cshim_dist_solver.restype = C.c_void_p
    assert lib.chase_hip_cshim_dist_solver(0)                       # visible from THIS thread
    deg, tol = C.c_int(20), C.c_double(1e-10)
    lib.pdchase_(C.byref(deg), C.byref(tol), C.c_char_p(b"R"), C.c_char_p(b"S"), C.c_char_p(b"C"))
    assert np.max(np.abs(ritzv[:nev] - (-N + 2.0 * np.arange(nev)))) < 1e-4      # Clement spectrum (1e-6 perturbation)
    assert np.max(O.residuals(H, ritzv[:nev], V[:, :nev])) < 1e-8
    flag = C.c_int(3)
    lib.pdchase_finalize_(C.byref(flag))
    assert flag.value == 0 and not lib.chase_hip_cshim_dist_solver(0)
    grid.close()


def test_c_interface_reinit_from_another_thread_replaces_the_solver(ctx):
    """init on thread A, init AGAIN on thread B without a finalize in between, solve on thread C: legal with the reference,
    whose static solver is simply replaced (chase_c_interface.cpp:905-1290) - the second problem is the one that is solved, from
    any thread (the advisor's finding of round 4: the second init used to land in a private slot of thread B)."""
    import ctypes as C
    import threading
    import numpy as np
    from chase_amd.capi import lib
    from chase_amd import dist as cd
    from oracle import chase_oracle as O
    assert lib.chase_hip_cshim_thread_ranks(0) == 0
    I = lambda v: C.byref(C.c_int(v))
    grid = cd.Grid(ctx, 1, 1, 0, transport="host", pg=None)
    probs = []
    for N, nev, nex in ((160, 12, 10), (220, 18, 12)):
        probs.append({"N": N, "nev": nev, "nex": nex, "H": np.asfortranarray(O.clement(N, False)),
                      "V": np.zeros((N, nev + nex), order="F"), "ritzv": np.zeros(nev + nex), "init": C.c_int(0)})

    def init(pr):
        lib.chase_hip_cshim_use_ctx(C.c_void_p(ctx.h.value), 0)
        lib.pdchase_init_hip_(I(pr["N"]), I(pr["nev"]), I(pr["nex"]), I(pr["N"]), I(pr["N"]), C.c_void_p(pr["H"].ctypes.data),
                              I(pr["N"]), C.c_void_p(pr["V"].ctypes.data), C.c_void_p(pr["ritzv"].ctypes.data),
                              C.c_void_p(grid.h.value), C.byref(pr["init"]))

    for pr in probs:                                   # thread A, then thread B
        t = threading.Thread(target=init, args=(pr,))
        t.start(); t.join()
        assert pr["init"].value == 1, lib.chase_hip_last_error()

    def solve():
        deg, tol = C.c_int(20), C.c_double(1e-10)
        lib.pdchase_(C.byref(deg), C.byref(tol), C.c_char_p(b"R"), C.c_char_p(b"S"), C.c_char_p(b"C"))

    t = threading.Thread(target=solve)                 # thread C
    t.start(); t.join()
    a, b = probs
    assert not np.any(a["ritzv"]) and not np.any(a["V"])                   # the replaced solver's buffers were never touched
    assert np.max(np.abs(b["ritzv"][:b["nev"]] - (-b["N"] + 2.0 * np.arange(b["nev"])))) < 1e-4
    assert np.max(O.residuals(b["H"], b["ritzv"][:b["nev"]], b["V"][:, :b["nev"]])) < 1e-8
    flag = C.c_int(3)
    lib.pdchase_finalize_(C.byref(flag))
    lib.chase_hip_cshim_dist_solver.restype = C.c_void_p
    assert flag.value == 0 and not lib.chase_hip_cshim_dist_solver(0)
    grid.close()


# ---- the shared-device transport (ranks = threads on ONE GPU, device-side collectives): what the full-size tests run on -------
def test_shared_device_transport_operators_and_exchanges():
    run_shared(4, S.scenario_hemm_kat)
    run_shared(4, S.scenario_p2p)
    run_shared(6, S.scenario_p2p)
    run_shared(4, S.scenario_ops, True, 16)
    run_shared(8, S.scenario_ops, False, 0)
    run_shared(6, S.scenario_sym_or_herm, True, 16)


def test_shared_device_transport_solves():
    run_shared(4, S.scenario_solve, 1001, 100, 60, False, 64, 20)
    run_shared(8, S.scenario_solve_counts, 600, 40, 24, True, 16, 20)
    run_shared(4, S.scenario_pseudo_solve, 0)
    run_shared(6, S.scenario_qr_fixtures, True, 16)
    run_shared(2, S.scenario_knob_switching, 640, 40, 24, True, 16, 20)


def test_shared_device_transport_releases_the_ranks_when_one_fails():
    """a rank that fails between collectives aborts the fabric: the others, waiting inside a device-side collective, come back
    with an error instead of waiting for the time-out"""
    import time
    import numpy as np
    from chase_amd.capi import lib, ChaseHipError, check
    from chase_amd import dist as cd

    def body(ctx, grid, comm):
        d = ctx.array(np.ones(1000))
        check(lib.chase_hip_grid_allreduce(grid.h, cd.COL, d.ptr, 1000, 0), "allreduce")
        assert np.all(d.download() == grid.nprow)
        if comm.rank == 1:
            raise ValueError("rank 1 gives up")
        check(lib.chase_hip_grid_allreduce(grid.h, cd.COL, d.ptr, 1000, 0), "allreduce")

    t = time.time()
    with pytest.raises(AssertionError, match="rank 1 gives up"):
        run_ranks(2, 1, body, transport="shared")
    assert time.time() - t < 60


def test_rayleigh_ritz_guard_catches_a_rank_whose_eigensolver_differs(monkeypatch):
    """one rank of a 2 x 2 grid returns other eigenvectors from heevd (fault injection): every rank notices by the 64-bit
    content hash and takes rank (0, 0)'s result; the solve ends like the undisturbed one"""
    monkeypatch.setenv("CHASE_HIP_RR_GUARD_FAULT", "3")
    run(4, S.scenario_rr_guard, 640, 40, 24, True, 16, 20)
