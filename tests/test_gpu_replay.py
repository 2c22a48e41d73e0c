"""Single-rank replay (chase_amd/replay.py, chase_amd/host/tape.hpp, chase_hip_grid_create_loopback): ONE rank of a grid driven
through the taped call sequence of a real solve with nothing on the other side.  What makes its time the real rank's time is
that it executes the real rank's operators - checked here launch for launch: the operator log (chase_hip_ctx_oplog: name and
shapes of every C-ABI operator, collectives and per-panel events included) of rank r in a REAL 2 x 2 / 2 x 1 / 3 x 2 solve (ranks as
threads, host transport) equals the operator log of the lone replayed rank r."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rank_threads import run_ranks  # noqa: E402
from chase_amd import dist as cd  # noqa: E402
from chase_amd.capi import lib, check, tape_mode, tape_get, tape_load  # noqa: E402
from oracle import chase_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def test_loopback_grid_keeps_ordering_and_moves_nothing(ctx):
    g = cd.Grid(ctx, 4, 2, 5, transport="loopback")
    assert (g.myrow, g.mycol) == (1, 1)
    assert g.transport_info() == (False, 1, 1)                   # not RCCL; nobody else in the communicators
    kind = cd.c_int()
    check(lib.chase_hip_grid_transport(g.h, cd.C.byref(kind), None, None), "transport")
    assert kind.value == 2
    assert lib.chase_hip_grid_group_active(g.h, cd.ROW) == 1 and lib.chase_hip_grid_group_active(g.h, cd.COL) == 1
    g.set_profiling(True)
    x = np.arange(1000.0)
    d = ctx.array(x)
    for grp in (cd.ROW, cd.COL):
        check(lib.chase_hip_grid_allreduce(g.h, grp, d.ptr, 1000, 1), "allreduce")
        check(lib.chase_hip_grid_event_record_on(g.h, grp, 3), "record")
        check(lib.chase_hip_grid_bcast(g.h, grp, d.ptr, 1000, 1, 0), "bcast")
    check(lib.chase_hip_grid_event_wait(g.h, 3), "wait")
    check(lib.chase_hip_grid_wait(g.h), "wait")
    assert np.array_equal(d.download().ravel(), x)
    v = cd.c_int(7)
    check(lib.chase_hip_grid_agree_max(g.h, cd.C.byref(v)), "agree_max")
    assert v.value == 7
    ms, waits = g.comm_exposed_ms()
    assert waits >= 3 and ms >= 0.0
    for n in (1, 2, 1, 2):                                       # switching the stream mapping between collectives
        g.set_comm_streams(n)
        assert g.comm_streams() == n
        check(lib.chase_hip_grid_allreduce(g.h, cd.COL, d.ptr, 1000, 1), "allreduce")
        check(lib.chase_hip_grid_allreduce(g.h, cd.ROW, d.ptr, 1000, 0), "allreduce")
    assert np.array_equal(d.download().ravel(), x)
    d.free()
    g.close()


def _real_rank(ctx, grid, comm, N, nev, nex, cplx, mb, deg, out):
    H = O.clement(N, cplx)
    rl, cl = cd.Layout(N, mb, grid.nprow), cd.Layout(N, mb, grid.npcol)
    dH = ctx.array(cd.local_block_of(H, rl, cl, grid.myrow, grid.mycol))
    s = cd.DistSolver(ctx, grid, dH, N, nev, nex, cplx, mb, mb)
    s.set(deg=deg)
    tape_mode(s, 1)
    comm.barrier()
    ctx.oplog(True)
    st = s.solve(trace=True)
    ctx.oplog(False)
    out[comm.rank] = {"oplog": ctx.oplog_lines(), "tape": tape_get(s), "stats": st, "trace": s.trace(),
                      "lam": s.ritzv[:nev].copy()}
    s.close()


@pytest.mark.parametrize("grid,N,nev,nex,cplx,mb,deg", [((2, 2), 600, 40, 24, True, 16, 20), ((2, 1), 500, 30, 20, False, 0, 16),
                                                       ((3, 2), 640, 36, 28, True, 32, 20)])
def test_replayed_rank_issues_the_real_ranks_launch_list(ctx, grid, N, nev, nex, cplx, mb, deg):
    nprow, npcol = grid
    real = {}
    run_ranks(nprow, npcol, _real_rank, N, nev, nex, cplx, mb, deg, real)
    # the recording is one and the same on every rank (the host outputs are agreed between the ranks)
    for r in range(1, nprow * npcol):
        assert np.array_equal(real[r]["tape"], real[0]["tape"])
    lam_exact = -N + 2.0 * np.arange(nev)
    assert np.max(np.abs(real[0]["lam"] - lam_exact)) < 1e-4      # (the test matrix carries a 1e-6 perturbation)
    H = O.clement(N, cplx)
    rl, cl = cd.Layout(N, mb, nprow), cd.Layout(N, mb, npcol)
    for r in range(nprow * npcol):
        g = cd.Grid(ctx, nprow, npcol, r, transport="loopback")
        dH = ctx.array(cd.local_block_of(H, rl, cl, g.myrow, g.mycol))
        s = cd.DistSolver(ctx, g, dH, N, nev, nex, cplx, mb, mb)
        s.set(deg=deg)
        tape_load(s, real[r]["tape"])
        tape_mode(s, 2)
        ctx.oplog(True)
        st = s.solve(trace=True)
        ctx.oplog(False)
        log = ctx.oplog_lines()
        # the driver took the recorded path ...
        assert st["iterations"] == real[r]["stats"]["iterations"] and st["filtered_vecs"] == real[r]["stats"]["filtered_vecs"]
        assert s.trace() == real[r]["trace"]
        assert s.get("tape_position") == s.get("tape_size") and s.get("tape_qr_mismatches") == 0
        # ... it was shown the recorded Ritz values although the lone rank computed something else ...
        assert np.array_equal(s.ritzv[:nev], real[r]["lam"])
        # ... and the lone rank executed the real rank's operators, launch for launch
        want = real[r]["oplog"]
        assert len(log) == len(want) and len(log) > 200, (len(log), len(want))
        diff = [(i, a, b) for i, (a, b) in enumerate(zip(log, want)) if a != b]
        assert not diff, diff[:5]
        assert any(l.startswith("allreduce") for l in log) and any(l.startswith("event_wait") for l in log)
        s.close()
        dH.free()
        g.close()


def _real_rank_pseudo(ctx, grid, comm, N, nev, nex, out):
    rl, cl = cd.Layout(N, 0, grid.nprow), cd.Layout(N, 0, grid.npcol)
    dH = cd.gen_bse_local(ctx, N, True, rl, cl, grid.myrow, grid.mycol, dmin=1.0, dmax=11.0, offdiag=1e-3)
    s = cd.DistPseudoSolver(ctx, grid, dH, N, nev, nex, True, 0, 0)
    s.set(device_rng=1, numlanczos=10, lanczositer=50)
    tape_mode(s, 1)
    comm.barrier()
    ctx.oplog(True)
    st = s.solve(trace=True)
    ctx.oplog(False)
    out[comm.rank] = {"oplog": ctx.oplog_lines(), "tape": tape_get(s), "stats": st, "trace": s.trace(),
                      "lam": s.ritzv[:nev].copy(), "resid": float(np.max(s.resid()[:nev]))}
    s.close()
    dH.free()


@pytest.mark.parametrize("grid", [(2, 2), (4, 2)])
def test_replayed_rank_of_the_pseudo_hermitian_solve_issues_the_real_ranks_launch_list(ctx, grid):
    """The same for chase::Solve_pseudo on the grid Impl (BASELINE configs[4]'s path: H^2 filter, K-conjugation, S-orthogonal QR,
    rayleighRitz_v2): the tape of a real 2 x 2 / 4 x 2 solve of a synthetic Bethe-Salpeter matrix replayed on every rank alone.  A lone
    rank's projected matrix Q^H S H Q is a partial sum and need not factorise: the replayed kernel then runs the dense core on the
    identity (HipImplExtras::set_replay_tolerant; counted, kept out of the operator log)."""
    nprow, npcol = grid
    N, nev, nex = 1024, 24, 16
    real = {}
    run_ranks(nprow, npcol, _real_rank_pseudo, N, nev, nex, real)
    for r in range(1, nprow * npcol):
        assert np.array_equal(real[r]["tape"], real[0]["tape"])
    assert real[0]["stats"]["locked"] >= nev and real[0]["resid"] <= 1e-10 and np.all(real[0]["lam"] > 0)
    rl, cl = cd.Layout(N, 0, nprow), cd.Layout(N, 0, npcol)
    tolerated = 0
    for r in range(nprow * npcol):
        g = cd.Grid(ctx, nprow, npcol, r, transport="loopback")
        dH = cd.gen_bse_local(ctx, N, True, rl, cl, g.myrow, g.mycol, dmin=1.0, dmax=11.0, offdiag=1e-3)
        s = cd.DistPseudoSolver(ctx, g, dH, N, nev, nex, True, 0, 0)
        s.set(device_rng=1, numlanczos=10, lanczositer=50)
        tape_load(s, real[r]["tape"])
        tape_mode(s, 2)
        ctx.oplog(True)
        st = s.solve(trace=True)
        ctx.oplog(False)
        log = ctx.oplog_lines()
        assert st["iterations"] == real[r]["stats"]["iterations"] and st["filtered_vecs"] == real[r]["stats"]["filtered_vecs"]
        assert s.trace() == real[r]["trace"]
        assert s.get("tape_position") == s.get("tape_size") and s.get("tape_qr_mismatches") == 0
        assert np.array_equal(s.ritzv[:nev], real[r]["lam"])
        tolerated += int(s.get("tape_tolerated"))
        want = real[r]["oplog"]
        assert len(log) == len(want) and len(log) > 200, (len(log), len(want))
        diff = [(i, a, b) for i, (a, b) in enumerate(zip(log, want)) if a != b]
        assert not diff, diff[:5]
        assert any(l.startswith("sendrecv") for l in log) or nprow == 1          # the K-conjugate exchange
        s.close()
        dH.free()
        g.close()
    print("projected matrices replaced by the identity over all replayed ranks:", tolerated)


def test_replay_through_bench_cli(tmp_path):
    """bench.py --replay-rank: records the tape of a real single-GPU solve (cfg1), replays rank 0 of 2x2 and 2x1"""
    tape = tmp_path / "tape.npz"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--replay-rank", "2x2,2x1", "--workload", "cfg1", "--tape", str(tape),
           "--oplog-out", str(tmp_path / "oplog_%g.txt")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["single_gpu"]["spectrum_check"]["ok"] and out["single_gpu"]["locked"] >= 100
    assert [r["grid"] for r in out["replays"]] == ["2x2", "2x1"]
    for r in out["replays"]:
        assert r["call_sequence_equals_recording"] and r["qr_variant_mismatches"] == 0
        assert r["local_shape_H"] == [2048, 4096 // int(r["grid"][-1])]
        assert r["T_rank_seconds"] > 0 and r["waits_on_communication_streams"] > 0
        assert os.path.getsize(tmp_path / f"oplog_{r['grid']}.txt") > 1000
    # a second call loads the tape instead of solving again; another rank of the grids, against MODELLED collectives (each holds
    # its communication stream and 8 RCCL-sized workgroups for 20 us + wire bytes / 20 GB/s), after the first-contact self-tuning
    # of the panel pipeline has run on the replayed rank (three trials)
    p = subprocess.run(cmd[:-2] + ["--replay-rank-index", "1", "--loopback-busbw", "20", "--loopback-wgs", "8", "--replay-autotune", "3"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "tape loaded" in p.stderr
    out2 = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    for r, r0 in zip(out2["replays"], out["replays"]):
        assert r["rank"] == 1 and r["call_sequence_equals_recording"] and r["qr_variant_mismatches"] == 0
        assert r["loopback_model"] == {"busbw_GBps": 20.0, "latency_us": 20.0, "touch": True, "workgroups": 8}
        assert r["exposed_ms_of_those_waits_with_nothing_on_the_wire"] > r0["exposed_ms_of_those_waits_with_nothing_on_the_wire"]
        a = r["autotune"]
        assert len(a["trials"]) == 3 and sum(t["kept"] for t in a["trials"]) == 1 and a["chosen"]["comm_streams"] == 1
