"""One rank of `bench.py --gpus N` (N > 1; also the 1x1-grid development runs): one process per GPU, 2D grid of the
reference (nprow x npcol with nprow >= npcol: 2x1, 2x2, 4x2; grid/mpiGrid2D.hpp), RCCL row/column all-reduces over xGMI,
the SAME workload as the single-GPU line (strong scaling).  torch.distributed (gloo over MASTER_ADDR) is used only for
bootstrap (unique-id exchange), barriers and the max / sum over ranks of the timing figures.

A rank whose communicator cannot be created exits non-zero: there is no fallback transport in a measured run (the
host-callback transport exists for tests on a single-GPU box: CHASE_HIP_TRANSPORT=host, labelled in config.workload)."""
import ctypes
import json
import os
import sys
import time

import numpy as np


def comm_probe(ctx, grid, dH, m_loc, n_loc, cplx, nevex, panel=256, reps=5):
    """Diagnostics for the multi-GPU runs (this session could only run them through size-1 communicators): what the filter's
    all-reduces cost on their own, and what one panel product of the pipelined HEMM costs alone, beside an all-reduce of the
    previous panel, and with the K-piece granularity switched off.  Every rank takes part; rank 0 reports."""
    from .capi import lib, check
    E = 2 if cplx else 1
    dt = np.complex128 if cplx else np.float64
    out = {"panel_cols": panel, "reps": reps}
    from .dist import ROW, COL                          # CHASE_HIP_ROW / CHASE_HIP_COL (include/chase_hip_grid.h)

    def timed(fn):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps

    for name, group, rows, size in (("col_group", COL, n_loc, grid.nprow), ("row_group", ROW, m_loc, grid.npcol)):
        if not lib.chase_hip_grid_group_active(grid.h, group):
            continue
        for label, cols in (("panel", panel), ("full_width", nevex)):
            buf = ctx.empty((rows, cols), dt)
            check(lib.chase_hip_fill_normal(ctx.h, int(cplx), rows, cols, buf.ptr, rows, 0, 0, rows, 7), "fill")
            ms = timed(lambda: check(lib.chase_hip_grid_allreduce(grid.h, group, buf.ptr, rows * cols * E, 0), "allreduce"))
            nbytes = rows * cols * E * 8
            out[f"{name}_{label}_allreduce"] = {"ranks": size, "bytes": nbytes, "ms": ms,
                                                 "algbw_GBps": nbytes / (ms * 1e-3) / 1e9,
                                                 "busbw_GBps": nbytes * 2.0 * (size - 1) / max(size, 1) / (ms * 1e-3) / 1e9}
            buf.free()
    # one panel of the column -> row product (W = H_loc^H V) alone / beside the all-reduce of another panel
    if lib.chase_hip_grid_group_active(grid.h, COL):
        V = ctx.empty((m_loc, panel), dt); W = ctx.empty((n_loc, panel), dt); X = ctx.empty((n_loc, panel), dt)
        check(lib.chase_hip_fill_normal(ctx.h, int(cplx), m_loc, panel, V.ptr, m_loc, 0, 0, m_loc, 8), "fill")
        check(lib.chase_hip_fill_normal(ctx.h, int(cplx), n_loc, panel, X.ptr, n_loc, 0, 0, n_loc, 9), "fill")
        lib.chase_hip_ctx_set_phase(ctx.h, 1)

        def gemm():
            ctx.gemm("C", n_loc, panel, m_loc, 0.5, dH.ptr, m_loc, V.ptr, m_loc, 0.0, W.ptr, n_loc, cplx)

        def both():
            check(lib.chase_hip_grid_allreduce(grid.h, COL, X.ptr, n_loc * panel * E, 1), "allreduce")
            gemm()
            check(lib.chase_hip_grid_wait(grid.h), "grid_wait")

        for rounds in (0, 4):
            lib.chase_hip_ctx_set_gemm_min_rounds(ctx.h, rounds)
            out[f"panel_gemm_alone_ms_rounds{rounds}"] = timed(gemm)
            out[f"panel_gemm_beside_allreduce_ms_rounds{rounds}"] = timed(both)
        lib.chase_hip_ctx_set_gemm_min_rounds(ctx.h, 0)
        lib.chase_hip_ctx_set_phase(ctx.h, 0)
        for a in (V, W, X):
            a.free()
    return out


def run_distributed(args):
    import torch
    import torch.distributed as dist
    from .capi import Context, gemm_counters
    from . import dist as cd
    import bench as B

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    dist.init_process_group("gloo")
    workload = args.workload or B.DEFAULT_WORKLOAD
    N, cplx, nev, nex = B.WORKLOADS[workload]
    if args.n:
        N = args.n
    nevex = nev + nex
    nprow, npcol = cd.grid_shape(world)
    myrow, mycol = cd.coords_of(rank, nprow)
    transport = os.environ.get("CHASE_HIP_TRANSPORT", "rccl")
    # health check BEFORE anybody enters ncclCommInitRank (which blocks until all members arrive): every rank must have
    # its device context; a rank without one makes ALL ranks leave with an error instead of leaving the others hanging
    ctx, err = None, ""
    try:
        ndev = torch.cuda.device_count()
        ctx = Context(local_rank % max(ndev, 1))
    except Exception as e:
        err = str(e)
    okflag = torch.tensor([1 if ctx is not None else 0], dtype=torch.int32)
    dist.all_reduce(okflag, op=dist.ReduceOp.MIN)
    if int(okflag[0]) == 0:
        print(f"bench rank {rank}: no usable device context ({err or 'another rank failed'})", file=sys.stderr, flush=True)
        dist.destroy_process_group()
        sys.exit(3)
    pg = cd.make_process_groups(nprow, npcol)
    try:
        grid = cd.Grid(ctx, nprow, npcol, rank, transport=transport, pg=pg)
    except Exception as e:
        # no silent fallback: a run that would measure PCIe + gloo instead of RCCL over xGMI must not print a value
        print(f"bench rank {rank}: {transport} grid creation failed: {e}", file=sys.stderr, flush=True)
        os._exit(4)
    is_rccl, rccl_row, rccl_col = grid.transport_info()
    # one line per rank for the audit of a multi-GPU run: which physical device, what the runtime was allowed to see, and
    # how many ranks RCCL itself counts in this rank's row / column communicator
    print(f"bench rank {rank}/{world}: grid ({myrow},{mycol}) of {nprow}x{npcol}, device bus id {ctx.bus_id()}, "
          f"bound to ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES', '<all>')} "
          f"({torch.cuda.device_count()} visible), transport {'rccl' if is_rccl else 'host'}, "
          f"ncclCommCount row {rccl_row} col {rccl_col}", file=sys.stderr, flush=True)
    grid.set_profiling(True)
    # RCCL prints a version banner through C stdio at communicator creation; push it out NOW on every rank so that the
    # JSON line rank 0 prints at the end is the last line of the job's stdout
    ctypes.CDLL(None).fflush(None)
    mb = nb = args.block_cyclic if args.block_cyclic >= 0 else B.DEFAULT_BLOCK_CYCLIC.get(workload, 0)
    rl, cl = cd.Layout(N, mb, nprow), cd.Layout(N, nb, npcol)
    pseudo = workload in B.PSEUDO_WORKLOADS
    if pseudo:
        dH = cd.gen_bse_local(ctx, N, cplx, rl, cl, myrow, mycol, **B.BSE_MATRIX)
        ctx.sync()
        s = cd.DistPseudoSolver(ctx, grid, dH, N, nev, nex, cplx, mb, nb)
        s.set(device_rng=1, numlanczos=10, lanczositer=50)      # the reference's BSE settings (5_bse_benchmark / BSE test)
    else:
        dH = cd.gen_clement_local(ctx, N, cplx, rl, cl, myrow, mycol, scale=B.MATRIX_SCALE / N, perturb=B.MATRIX_PERTURB)
        ctx.sync()
        s = cd.DistSolver(ctx, grid, dH, N, nev, nex, cplx, mb, nb)
        s.set(device_rng=1)

    def snapshot():
        model, execd, calls = gemm_counters(ctx, 1)
        exposed_ms, waits = grid.comm_exposed_ms()
        return {"filter_ms": s.get("filter_ms"), "hemm_calls": s.get("hemm_calls"), "reused": s.get("hemm_reused_vecs"),
                "model": model, "exec": execd, "gemms": calls, "exposed_ms": exposed_ms, "waits": waits}

    timer = B.StepTimer(args.steps, args.warmup, ctx.sync, dist.barrier, snapshot)
    complete, last = B.run_timed_solves(s, timer, nev, lambda: (s.ritzv[:nev].copy(), s.resid()[:nev].copy()))
    wall_loc = timer.t1 - timer.t0
    # MAX over ranks of the wall time, the filter time and the exposed communication; SUM of the kernel-side flop books
    t = torch.tensor([wall_loc, timer.diff("filter_ms") * 1e-3, timer.diff("exposed_ms")], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall, filt_s, exposed_ms = float(t[0]), float(t[1]), float(t[2])
    f = torch.tensor([timer.diff("model"), timer.diff("exec")], dtype=torch.float64)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    model_flops, exec_flops = float(f[0]), float(f[1])
    calls = int(timer.diff("hemm_calls"))
    reused = int(timer.diff("reused"))
    hemm_vecs = timer.filtered_timed - reused
    F = 4 if cplx else 1
    if not pseudo:
        formula = 2.0 * F * N * N * hemm_vecs                 # whole-job flops of the reference's model
        assert abs(model_flops - formula) <= 1e-9 * formula, (model_flops, formula)
    gflops = model_flops / filt_s / 1e9
    lam, resid = last
    spec = None if pseudo else B.spectrum_check(lam, N, nev)
    # independent residuals of the last solve's eigenvectors (fresh four-product H V, redistribution, all-reduce over the
    # row group: mpi/residuals.hpp:61-107 as it stands), outside the timed region; collective
    resid_re = s.recompute_residuals(nev, lam)
    ok = bool(np.max(resid) < 1e-8 and np.max(resid_re) < 1e-8 and (spec is None or spec["ok"]))
    st = complete[-1]
    solve_s = float(np.mean([c["t_all"] for c in complete]))
    tot = snapshot()
    tw = torch.tensor([tot["model"], tot["exec"]], dtype=torch.float64)
    dist.all_reduce(tw, op=dist.ReduceOp.SUM)
    tm = torch.tensor([tot["filter_ms"]], dtype=torch.float64)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    tot["model"], tot["exec"], tot["filter_ms"] = float(tw[0]), float(tw[1]), float(tm[0])
    probe = None
    # (the host-callback test transport runs it only on request: CHASE_HIP_PROBE_HOST=1, to exercise the multi-rank code path)
    if (is_rccl or os.environ.get("CHASE_HIP_PROBE_HOST") == "1") and not pseudo and not getattr(args, "no_probe", False):
        dist.barrier()
        probe = comm_probe(ctx, grid, dH, rl.count(myrow), cl.count(mycol), cplx, nevex)
    out = None
    if rank == 0:
        out = {
            "metric": "chebyshev_filter_hemm_gflops", "value": gflops, "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "complex f64" if cplx else "f64", "data": "synthetic",
            "config": {"workload": f"{workload}: ChASE solve, "
                                   + ("synthetic Bethe-Salpeter pseudo-Hermitian (Solve_pseudo, H^2 filter)" if pseudo
                                      else "perturbed Clement-type Hermitian (x100/N)") + f" N={N} "
                                   f"{'complex' if cplx else 'real'} fp64, nev={nev} nex={nex}, tol 1e-10, deg 20 opt, "
                                   f"{nprow}x{npcol} {'block-cyclic nb=%d' % nb if nb else 'block'} grid, "
                                   + ("RCCL over xGMI" if is_rccl else "host-callback (gloo) TEST transport - not a measurement of RCCL")
                                   + "; step = one outer iteration (filter+QR+RR+residuals+locking), solves back to back",
                       "N": N, "nev": nev, "nex": nex, "grid": f"{nprow}x{npcol}", "step": "outer iteration",
                       "transport": "rccl" if is_rccl else "host"},
            "eigenpairs_per_sec": nev / solve_s, "solve_seconds": solve_s, "complete_solves": len(complete),
            "pct_fp64_mfma_peak": 100.0 * exec_flops / filt_s / 1e12 / world / B.FP64_MFMA_PEAK_TFLOPS,
            "converged": ok, "max_resid": float(np.max(resid)), "max_resid_recomputed": float(np.max(resid_re)),
            "spectrum_check": spec,
            "iterations_per_solve": st["iterations"], "filtered_vecs_per_solve": st["filtered_vecs"],
            "timed": {"filtered_vecs": timer.filtered_timed, "hemm_vecs": hemm_vecs, "first_step_vecs_from_rr": reused,
                      "filter_seconds_device": filt_s, "wall_seconds": wall,
                      "iterations": [{"solve": a, "iteration": b, "filtered_vecs": c, "seconds": d}
                                     for a, b, c, d in timer.per_iter]},
            "phase_seconds_last_complete_solve": {k: st[k] for k in B.PHASES},
            # diagnosis of the multi-GPU run: what RCCL itself reports, and how long the compute stream sat waiting for a
            # collective with nothing else to run (bracketing events around every wait on the communication stream)
            "ranks_seen_by_rccl": {"row_communicator": rccl_row, "col_communicator": rccl_col,
                                   "grid": rccl_row * rccl_col if is_rccl else None},
            "comm_exposed_ms": exposed_ms, "comm_exposed_frac_of_wall": exposed_ms * 1e-3 / wall,
            "comm_waits": int(timer.diff("waits")),
            "roofline": B.roofline_object(model_flops, exec_flops, filt_s, calls, world,
                                          "; filter time includes the row/column all-reduces"),
        }
        out["roofline"]["whole_run"] = B.whole_run_object(tot, world)
        out["comm_probe"] = probe
    s.close()
    grid.close()
    del dH
    ctx.close()
    ctypes.CDLL(None).fflush(None)
    dist.barrier()
    dist.destroy_process_group()
    return out
