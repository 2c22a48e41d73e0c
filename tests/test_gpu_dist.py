"""Distributed Impl (pChaseHip + grid collectives) on ONE GPU: N ranks share the device and exchange through the
host-callback transport (gloo), so every line of the distributed C++ path except ncclAllReduce itself runs here.
The RCCL transport is exercised on a 1x1 grid (communicator-free) and by bench.py --gpus N on the multi-GPU node."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PORT = [29611]


def run_ranks(nranks, transport, *args, timeout=600):
    _PORT[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_PORT[0]),
           os.path.join(ROOT, "tests", "dist_worker.py"), transport, *map(str, args)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0 and "DIST_WORKER_OK" in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])


@pytest.mark.parametrize("nranks", [2, 4])
def test_hemm_known_answer(nranks):
    run_ranks(nranks, "host", "hemm_kat")


@pytest.mark.parametrize("nranks,typ,mb", [(4, "d", 0), (4, "z", 0), (4, "z", 16), (2, "d", 32), (6, "z", 0)])
def test_operators_vs_oracle(nranks, typ, mb):
    run_ranks(nranks, "host", "ops", typ, mb)


def test_solve_block_2x2_n256_complex():
    run_ranks(4, "host", "solve", 256, 24, 16, "z", 0, 16)


def test_solve_blockcyclic_2x2_n1001_nb64():
    # the reference's distributed integration test: N = 1001, nev = 100, nex = 60, nb = 64 on a 2 x 2 grid
    run_ranks(4, "host", "solve", 1001, 100, 60, "d", 64, 20)


def test_solve_rccl_transport_single_rank():
    run_ranks(1, "rccl", "solve", 256, 24, 16, "z", 0, 16)


def test_solve_rccl_forced_through_size1_communicators(monkeypatch):
    """CHASE_HIP_RCCL_FORCE routes the size-1 row/column groups through real RCCL communicators: ncclCommInitRank,
    ncclAllReduce, ncclBroadcast, the communication stream, the per-panel events of the pipelined HEMM."""
    monkeypatch.setenv("CHASE_HIP_RCCL_FORCE", "1")
    run_ranks(1, "rccl", "solve", 1001, 100, 60, "z", 64, 20)
    run_ranks(1, "rccl", "ops", "d", 0)


# ---- distributed pseudo-Hermitian (BSE) Impl: BASELINE config 5 / SURVEY.md §8 A11 ------------------------------------
@pytest.mark.parametrize("nranks,mb", [(2, 0), (4, 0), (4, 16), (6, 0)])
def test_pseudo_operators_vs_oracle(nranks, mb):
    run_ranks(nranks, "host", "pseudo_ops", mb)


@pytest.mark.parametrize("nranks,mb", [(4, 0), (4, 32), (2, 0)])
def test_pseudo_solve_bse_fixture(nranks, mb):
    run_ranks(nranks, "host", "pseudo_solve", mb)


def test_pseudo_solve_rccl_single_rank():
    run_ranks(1, "rccl", "pseudo_solve", 0)


def test_pseudo_rccl_forced_through_size1_communicators(monkeypatch):
    """the panel-pipelined pseudo-Hermitian filter (events, communication stream, real ncclAllReduce calls)"""
    monkeypatch.setenv("CHASE_HIP_RCCL_FORCE", "1")
    run_ranks(1, "rccl", "pseudo_solve", 0)
    run_ranks(1, "rccl", "pseudo_ops", 0)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "z", 0), (6, "d", 16), (1, "z", 0)])
def test_distributed_symmetry_check(nranks, typ, mb):
    run_ranks(nranks, "host", "symcheck", typ, mb)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "z", 0), (2, "d", 0), (6, "z", 0), (8, "d", 0), (4, "z", 8), (6, "d", 16), (1, "z", 0)])
def test_distributed_qr_on_reference_fixtures(nranks, typ, mb):
    """CholQR variants and the DISTRIBUTED Householder (panel factorisation over the row-distributed block: pivots cross
    rank boundaries with 6 and 8 ranks, block-cyclic rows with mb > 0) on the reference's conditioned fixtures"""
    run_ranks(nranks, "host", "qr_fixtures", typ, mb)


def test_solve_blockcyclic_4x2_eight_ranks():
    """the 8-GPU grid shape of BASELINE configs[3] (4 x 2, block-cyclic nb = 64), eight ranks sharing one GPU"""
    run_ranks(8, "host", "solve", 1024, 100, 60, "z", 64, 20, timeout=900)


def test_pseudo_solve_4x2_eight_ranks():
    run_ranks(8, "host", "pseudo_solve", 0, timeout=900)


def test_distributed_run_reproduces_the_reference_example_run():
    """examples/1_hello_world measured with the actual reference (pChASECPU, 2 x 2, block-cyclic nb = 64): 6 iterations,
    13 310 filtered vectors (BASELINE.md cross-check table)"""
    run_ranks(4, "host", "refcounts")


@pytest.mark.parametrize("nranks,mb", [(4, 0), (2, 16)])
def test_pseudo_solve_real_fixture(nranks, mb):
    run_ranks(nranks, "host", "pseudo_solve_real", mb)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "d", 0), (4, "z", 0), (4, "z", 8), (2, "d", 16)])
def test_reference_distributed_kernel_tests(nranks, typ, mb):
    """tests/linalg/internal/mpi/{rayleighRitz,residuals,lanczos}.cpp (2 x 2 grid in the reference) through the grid Impl"""
    run_ranks(nranks, "host", "reference_units", typ, mb)


@pytest.mark.parametrize("nranks,typ,mb", [(4, "d", 0), (4, "z", 32), (2, "z", 0)])
def test_distributed_c_entry_points(nranks, typ, mb):
    """p?chase_init[_blockcyclic]_hip_ / p?chase_ / p?chase_get_eigenpairs_ / p?chase_wrtHam_ / readHam_ / finalize_"""
    run_ranks(nranks, "host", "cshim", typ, mb)


@pytest.mark.parametrize("nranks", [4, 6])
def test_grid_sendrecv_and_exact_agree_max(nranks):
    run_ranks(nranks, "host", "p2p")


def test_reference_mpi_signatures_on_one_rank():
    """libchase_hip_mpi.so: the reference's exact MPI_Comm* entry points (pzchase_init_blockcyclic_ ..., built when mpi.h is
    found).  One MPI rank (singleton MPI_Init, no launcher) on a 1 x 1 grid: communicator split, id broadcast, grid and
    context creation, solve, finalize releasing both."""
    mpi_lib = os.path.join(ROOT, "chase_amd", "lib", "libchase_hip_mpi.so")
    if not os.path.exists(mpi_lib) or not os.path.exists("/opt/conda/lib/libmpi.so.12"):
        pytest.skip("MPI front end not built (no mpi.h / libmpi on this box)")
    code = """
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
from chase_amd.capi import lib
from oracle import chase_oracle as O
mpi = C.CDLL("/opt/conda/lib/libmpi.so.12", mode=C.RTLD_GLOBAL)
assert mpi.MPI_Init(None, None) == 0
front = C.CDLL(%r)
world = C.c_int(0x44000000)                      # MPICH's MPI_COMM_WORLD handle
N, nev, nex, nb = 300, 24, 16, 32
H = O.clement(N, True)
V = np.zeros((N, nev + nex), dtype=complex, order="F"); ritzv = np.zeros(nev + nex)
I = lambda v: C.byref(C.c_int(v))
init = C.c_int(0)
front.pzchase_init_blockcyclic_(I(N), I(nev), I(nex), I(nb), I(nb), C.c_void_p(H.ctypes.data), I(N), C.c_void_p(V.ctypes.data),
                                C.c_void_p(ritzv.ctypes.data), I(1), I(1), C.c_char_p(b"C"), I(0), I(0), C.byref(world), C.byref(init))
assert init.value == 1, lib.chase_hip_last_error()
deg, tol = C.c_int(20), C.c_double(1e-10)
lib.pzchase_(C.byref(deg), C.byref(tol), C.c_char_p(b"R"), C.c_char_p(b"S"), C.c_char_p(b"C"))
k = O.OracleCPU(H, nev, nex); O.solve(k)
assert np.max(np.abs(ritzv[:nev] - k.ritzv[:nev])) < 1e-8
assert np.max(O.residuals(H, ritzv[:nev], V[:, :nev])) < 1e-8
flag = C.c_int(5); lib.pzchase_finalize_(C.byref(flag)); assert flag.value == 0
# the block-layout entry point rejects a local shape that does not match the layout
front.pzchase_init_(I(N), I(nev), I(nex), I(N - 1), I(N), C.c_void_p(H.ctypes.data), I(N), C.c_void_p(V.ctypes.data),
                    C.c_void_p(ritzv.ctypes.data), I(1), I(1), C.c_char_p(b"R"), C.byref(world), C.byref(init))
assert init.value == 0
mpi.MPI_Finalize()
print("MPI_FRONT_OK")
""" % (ROOT, mpi_lib)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0 and "MPI_FRONT_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


def test_c_mpi_application_example(tmp_path):
    """examples/c_dist_mpi.c: an MPI program in plain C on the reference's distributed entry points (pzchase_init_, pzchase_,
    pzchase_finalize_), built with gcc against libchase_hip_mpi.so + libchase_hip.so and run as one MPI rank (one GPU here)."""
    import shutil
    mpi_inc, mpi_lib = "/opt/conda/include", "/opt/conda/lib"
    lib = os.path.join(ROOT, "chase_amd", "lib")
    if not (os.path.exists(os.path.join(lib, "libchase_hip_mpi.so")) and os.path.exists(os.path.join(mpi_inc, "mpi.h"))
            and shutil.which("gcc")):
        pytest.skip("no MPI / gcc on this box")
    exe = str(tmp_path / "c_dist_mpi")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-I" + os.path.join(ROOT, "include"), "-I" + mpi_inc,
                    os.path.join(ROOT, "examples", "c_dist_mpi.c"), "-L" + lib, "-lchase_hip_mpi", "-lchase_hip",
                    os.path.join(mpi_lib, "libmpi.so"), "-Wl,--allow-shlib-undefined", "-Wl,--enable-new-dtags",
                    "-Wl,-rpath," + lib, "-Wl,-rpath," + mpi_lib, "-lm", "-o", exe],
                   check=True)
    p = subprocess.run([exe, "600"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0 and "-> OK" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
