"""The ranks of `bench.py --gpus N` (N > 1; also the 1x1-grid development runs): 2D grid of the reference (nprow x npcol with
nprow >= npcol: 2x1, 2x2, 4x2; grid/mpiGrid2D.hpp), RCCL row/column all-reduces over xGMI, the SAME workload as the
single-GPU line (strong scaling).  Two forms: one PROCESS per GPU (run_distributed: torch.distributed / gloo over
MASTER_ADDR only for bootstrap - unique-id exchange -, barriers and the max / sum over ranks of the timing figures) and one
THREAD per GPU inside one process (run_threads).  Either way a run first PROVES its transport (transport_proof: bus
bandwidth of a 256 MB all-reduce per communicator, in the JSON line); a probe child of `--ranks auto` exits with status 5 when
a communicator is not on xGMI (the next mode is tried), a measured run marks its line scaling_valid = false.

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


def transport_proof(ctx, grid, comm, nbytes=256 << 20, reps=3):
    """Before any solve: one 256 MB all-reduce per communicator, timed on its own.  A size >= 2 RCCL group that moves less
    than MIN_BUSBW_GBPS is not running over xGMI peer-to-peer (host-staged rings over PCIe reach 10-25 GB/s, one xGMI link
    ~50 GB/s per direction and the 7 links of a device several times that): a PROBE child then exits with status 5 (`--ranks
    auto` moves on to the next mode), a measured run carries on and marks its line scaling_valid = false.  Returns
    (record, ok) identical on every rank."""
    from .capi import lib, check
    from .dist import ROW, COL
    rec, ok = {"bytes": nbytes, "reps": reps, "min_busbw_GBps_required": MIN_BUSBW_GBPS}, True
    is_rccl = grid.transport_info()[0]
    count = nbytes // 8
    for name, group, size in (("row_group", ROW, grid.npcol), ("col_group", COL, grid.nprow)):
        if not lib.chase_hip_grid_group_active(grid.h, group):
            continue
        buf = ctx.empty((count,), np.float64)
        check(lib.chase_hip_memset(ctx.h, buf.ptr, 0, nbytes), "memset")
        check(lib.chase_hip_grid_allreduce(grid.h, group, buf.ptr, count, 0), "allreduce")      # connections, first touch
        ctx.sync()
        comm.barrier()
        ctx.timer_start()
        for _ in range(reps):
            check(lib.chase_hip_grid_allreduce(grid.h, group, buf.ptr, count, 0), "allreduce")
        ms = ctx.timer_stop() / reps
        buf.free()
        ms = comm.allreduce_max([ms])[0]
        busbw = nbytes * 2.0 * (size - 1) / size / (ms * 1e-3) / 1e9
        rec[name] = {"ranks": size, "ms": ms, "algbw_GBps": nbytes / (ms * 1e-3) / 1e9, "busbw_GBps": busbw}
        if is_rccl and size >= 2 and busbw < MIN_BUSBW_GBPS:
            ok = False
    # what a SMALL synchronous collective costs (64 doubles: Lanczos scalars, agreement collectives, the residual sums): per
    # communicator and alternating between the two - over RCCL's socket transport the alternating pattern on two communication
    # streams costs 20x the single-stream figure (profiles/r05_socket_rccl_streams.txt); reported, never part of `ok`
    import time
    small = ctx.empty((64,), np.float64)
    check(lib.chase_hip_memset(ctx.h, small.ptr, 0, 64 * 8), "memset")
    groups = [g for g in (COL, ROW) if lib.chase_hip_grid_group_active(grid.h, g)]
    lat = {}
    for name, seq in (("col_group", [COL]), ("row_group", [ROW]), ("alternating", [COL, ROW])):
        if any(g not in groups for g in seq):
            continue
        for g in seq:
            check(lib.chase_hip_grid_allreduce(grid.h, g, small.ptr, 64, 0), "allreduce")
        ctx.sync()
        comm.barrier()
        t = time.perf_counter()
        n = 50
        for _ in range(n):
            for g in seq:
                check(lib.chase_hip_grid_allreduce(grid.h, g, small.ptr, 64, 0), "allreduce")
        ctx.sync()
        lat[name] = comm.allreduce_max([(time.perf_counter() - t) / (n * len(seq)) * 1e6])[0]
    small.free()
    rec["small_allreduce_latency_us"] = dict(lat, comm_streams=grid.comm_streams())
    rec["ok"] = ok
    return rec, ok


# the bar sits between what a host-staged ring over PCIe reaches (10-25 GB/s) and ONE xGMI link (~50 GB/s per direction: a
# 2-rank row group of the 4 x 2 grid rides a single link, and the timed repetitions include the host-side event round trip)
MIN_BUSBW_GBPS = float(os.environ.get("CHASE_HIP_MIN_BUSBW_GBPS", "30"))
EXIT_TRANSPORT_TOO_SLOW = 5


class TransportTooSlow(RuntimeError):
    pass


def run_rank(args, comm, ctx, grid, mode):
    """One rank of the multi-GPU bench on an established context + grid; `comm` carries bootstrap, barriers and the max / sum
    over ranks of the timing figures (chase_amd.rank_threads.RankComm or GlooComm).  Rank 0 returns the result record."""
    from .capi import gemm_counters
    from . import dist as cd
    import bench as B

    rank, world = comm.rank, comm.world
    workload = args.workload or B.DEFAULT_WORKLOAD
    N, cplx, nev, nex = B.WORKLOADS[workload]
    if args.n:
        N = args.n
    nevex = nev + nex
    nprow, npcol = grid.nprow, grid.npcol
    myrow, mycol = grid.myrow, grid.mycol
    is_rccl, rccl_row, rccl_col = grid.transport_info()
    # one line per rank for the audit of a multi-GPU run: which physical device, what the runtime was allowed to see, and
    # how many ranks RCCL itself counts in this rank's row / column communicator
    print(f"bench rank {rank}/{world} [{mode}]: grid ({myrow},{mycol}) of {nprow}x{npcol}, ordinal {ctx.device}, device bus id "
          f"{ctx.bus_id()}, ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES', '<all>')}, transport "
          f"{'rccl' if is_rccl else grid.transport}, ncclCommCount row {rccl_row} col {rccl_col}", file=sys.stderr, flush=True)
    grid.set_profiling(True)
    # RCCL prints a version banner through C stdio at communicator creation; push it out NOW on every rank so that the
    # JSON line rank 0 prints at the end is the last line of the job's stdout
    ctypes.CDLL(None).fflush(None)
    proof, proof_ok = transport_proof(ctx, grid, comm)
    if rank == 0:
        print("bench: transport proof " + json.dumps(proof), file=sys.stderr, flush=True)
    if not proof_ok and rank == 0:
        # Round 5 (the advisor's finding): the bar was never calibrated on an xGMI box, so a slow communicator no longer ends
        # the MEASURED run without a line - the probe children of `--ranks auto` have already walked bound -> unbound ->
        # threads looking for a mode that passes; whatever mode runs now solves, and its line says transport_proof.ok = false
        # and scaling_valid = false so that nobody reads a PCIe-staged number as an xGMI one
        print("bench: WARNING: a communicator moves less than %.0f GB/s (bus bandwidth of a 256 MB all-reduce) in mode '%s': "
              "the line below is marked scaling_valid = false - %s" % (MIN_BUSBW_GBPS, mode, json.dumps(proof)),
              file=sys.stderr, flush=True)
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
    if os.environ.get("CHASE_HIP_PIPELINE") == "0":
        s.set(pipeline=0)            # development: every all-reduce waited for where it is issued (what the overlap is worth)
    # first contact with the hardware: the knobs of the panel pipeline are measured on identical full-width filter steps and
    # the best setting is locked BEFORE the first solve (chase_amd/autotune.py); untimed, a few seconds
    tuned = None
    if (world > 1 and not getattr(args, "no_autotune", False)
            and (is_rccl or os.environ.get("CHASE_HIP_AUTOTUNE_HOST") == "1")):
        from .autotune import first_contact
        tuned = first_contact(s, ctx, grid, comm, nevex, budget=int(os.environ.get("CHASE_HIP_AUTOTUNE_TRIALS", "5")),
                              log=(lambda m: print("bench: " + m, file=sys.stderr, flush=True)) if rank == 0 else None)
        s.set(reset_counters=1)
        for ph in range(4):
            gemm_counters(ctx, ph, reset=True)
        grid.comm_exposed_ms(reset=True)

    def snapshot():
        model, execd, calls = gemm_counters(ctx, 1)
        exposed_ms, waits = grid.comm_exposed_ms()
        return {"filter_ms": s.get("filter_ms"), "hemm_calls": s.get("hemm_calls"), "reused": s.get("hemm_reused_vecs"),
                "model": model, "exec": execd, "gemms": calls, "exposed_ms": exposed_ms, "waits": waits}

    timer = B.StepTimer(args.steps, args.warmup, ctx.sync, comm.barrier, snapshot, progress=(rank == 0))
    complete, last = B.run_timed_solves(s, timer, nev, lambda: (s.ritzv[:nev].copy(), s.resid()[:nev].copy()))
    wall_loc = timer.t1 - timer.t0
    # MAX over ranks of the wall time, the filter time and the exposed communication; SUM of the kernel-side flop books
    wall, filt_s, exposed_ms = comm.allreduce_max([wall_loc, timer.diff("filter_ms") * 1e-3, timer.diff("exposed_ms")])
    model_flops, exec_flops = comm.allreduce_sum([timer.diff("model"), timer.diff("exec")])
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
    tol = s.get("tol")
    ok = B.converged_ok(lam, resid, resid_re, tol, spec)
    st = complete[-1]
    solve_s = float(np.mean([c["t_all"] for c in complete]))
    tot = snapshot()
    tot["model"], tot["exec"] = comm.allreduce_sum([tot["model"], tot["exec"]])
    tot["filter_ms"] = comm.allreduce_max([tot["filter_ms"]])[0]
    probe = None
    # (the host-callback test transport runs it only on request: CHASE_HIP_PROBE_HOST=1, to exercise the multi-rank code path)
    if (is_rccl or os.environ.get("CHASE_HIP_PROBE_HOST") == "1") and not pseudo and not getattr(args, "no_probe", False):
        comm.barrier()
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
                                   + (("RCCL over xGMI" if os.environ.get("CHASE_BENCH_FAKE_HOSTS") != "1" else
                                       "RCCL over its SOCKET transport between rank processes sharing one GPU (NCCL_HOSTID per rank): "
                                       "a functional rehearsal, not a measurement") if is_rccl
                                      else ("shared-device TEST transport (rank threads on ONE GPU, device-side collectives) - not a "
                                            "measurement of RCCL" if grid.transport == "shared"
                                            else "host-callback TEST transport - not a measurement of RCCL"))
                                   + f", ranks = {mode}"
                                   + "; step = one outer iteration (filter+QR+RR+residuals+locking), solves back to back",
                       "N": N, "nev": nev, "nex": nex, "grid": f"{nprow}x{npcol}", "step": "outer iteration",
                       "transport": "rccl" if is_rccl else grid.transport, "ranks": mode},
            "eigenpairs_per_sec": nev / solve_s, "solve_seconds": solve_s, "complete_solves": len(complete),
            "pct_fp64_mfma_peak": 100.0 * exec_flops / filt_s / 1e12 / world / B.FP64_MFMA_PEAK_TFLOPS,
            "converged": ok, "max_resid": float(np.max(resid)), "max_resid_recomputed": float(np.max(resid_re)),
            "residuals_rechecked_on_the_tolerance": int(s.get("resd_rechecked")),
            "spectrum_check": spec,
            "iterations_per_solve": st["iterations"], "filtered_vecs_per_solve": st["filtered_vecs"],
            "timed": {"filtered_vecs": timer.filtered_timed, "hemm_vecs": hemm_vecs, "first_step_vecs_from_rr": reused,
                      "filter_seconds_device": filt_s, "wall_seconds": wall,
                      "iterations": [{"solve": a, "iteration": b, "filtered_vecs": c, "seconds": d}
                                     for a, b, c, d in timer.per_iter]},
            "phase_seconds_last_complete_solve": {k: st[k] for k in B.PHASES},
            # diagnosis of the multi-GPU run: what RCCL itself reports, what the communicators move on their own, and how
            # long the compute stream sat waiting for a collective with nothing else to run (bracketing events around every
            # wait on the communication stream)
            "ranks_seen_by_rccl": {"row_communicator": rccl_row, "col_communicator": rccl_col,
                                   "grid": rccl_row * rccl_col if is_rccl else None},
            "transport_proof": proof,
            # a scaling number only when RCCL really ran between devices: not on a test transport, not over fake hosts' sockets
            "scaling_valid": bool(proof_ok and ok and is_rccl and os.environ.get("CHASE_BENCH_FAKE_HOSTS") != "1"),
            "comm_exposed_ms": exposed_ms, "comm_exposed_frac_of_wall": exposed_ms * 1e-3 / wall,
            "comm_waits": int(timer.diff("waits")),
            "roofline": B.roofline_object(model_flops, exec_flops, filt_s, calls, world,
                                          "; filter time includes the row/column all-reduces"),
        }
        out["roofline"]["whole_run"] = B.whole_run_object(tot, world)
        out["comm_probe"] = probe
        out["autotune"] = tuned
    s.close()
    del dH
    return out


def run_distributed(args, probe_only=False):
    """Process-per-GPU form (the contract's launch: torch.distributed.run, or bench.py's own spawn_ranks): this process is
    one rank; gloo over MASTER_ADDR carries the bootstrap and the barriers."""
    import torch
    import torch.distributed as dist
    from .capi import Context
    from . import dist as cd
    from .rank_threads import GlooComm

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    # RCCL watchdog of grid.hip: a collective that has not finished this long after it was issued ends the rank with an error
    # (the library's default, 600 s, is the whole budget of a driver-run bench; the longest legitimate age is one filter call's
    # queued work: ~40 s at two GPUs)
    os.environ.setdefault("CHASE_HIP_FABRIC_TIMEOUT_S", "240")
    dist.init_process_group("gloo")
    comm = GlooComm()
    nprow, npcol = cd.grid_shape(world)
    transport = os.environ.get("CHASE_HIP_TRANSPORT", "rccl")
    # health check BEFORE anybody enters ncclCommInitRank (which blocks until all members arrive): every rank must have
    # its device context; a rank without one makes ALL ranks leave with an error instead of leaving the others hanging
    ctx, err = None, ""
    try:
        ndev = torch.cuda.device_count()
        ctx = Context(local_rank % max(ndev, 1))
    except Exception as e:
        err = str(e)
    if comm.allreduce_min([1 if ctx is not None else 0])[0] == 0:
        print(f"bench rank {rank}: no usable device context ({err or 'another rank failed'})", file=sys.stderr, flush=True)
        dist.destroy_process_group()
        sys.exit(3)
    try:
        pg = cd.make_process_groups(nprow, npcol) if transport != "rccl" else comm
        grid = cd.Grid(ctx, nprow, npcol, rank, transport=transport, pg=pg)
    except Exception as e:
        # no silent fallback: a run that would measure PCIe + gloo instead of RCCL over xGMI must not print a value
        print(f"bench rank {rank}: {transport} grid creation failed: {e}", file=sys.stderr, flush=True)
        os._exit(4)
    mode = "processes" + (", all devices visible" if "CHASE_HIP_BOUND_DEVICE" not in os.environ else ", one visible device each")
    if probe_only:
        # child of bench.py's mode selection: context, communicators and the transport proof, then out - status 0 = usable
        proof, ok = transport_proof(ctx, grid, comm)
        if rank == 0:
            print(f"bench probe [{mode}]: " + json.dumps(proof), file=sys.stderr, flush=True)
        grid.close()
        ctx.close()
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
        dist.destroy_process_group()
        os._exit(0 if ok else EXIT_TRANSPORT_TOO_SLOW)
    try:
        out = run_rank(args, comm, ctx, grid, mode)
    except TransportTooSlow as e:
        print(f"bench rank {rank}: {e}", file=sys.stderr, flush=True)
        ctypes.CDLL(None).fflush(None)
        os._exit(EXIT_TRANSPORT_TOO_SLOW)
    grid.close()
    ctx.close()
    ctypes.CDLL(None).fflush(None)
    dist.barrier()
    dist.destroy_process_group()
    return out


def run_threads(args, nranks):
    """`bench.py --gpus N --ranks threads`: ONE process, one thread per GPU.  Thread r opens device r (r modulo
    the visible devices, so that the launcher logic can be rehearsed with several threads on one GPU over the host
    transport), ncclCommInitRank runs from the N threads (ctypes drops the GIL around every library call), the bootstrap
    and the barriers stay inside the process.  No visibility tricks, no IPC handles, one process per card."""
    import torch
    from . import dist as cd
    from .rank_threads import run_ranks

    nprow, npcol = cd.grid_shape(nranks)
    transport = os.environ.get("CHASE_HIP_TRANSPORT", "rccl")
    os.environ.setdefault("CHASE_HIP_FABRIC_TIMEOUT_S", "240")      # (see run_distributed)
    ndev = max(torch.cuda.device_count(), 1)
    if transport == "rccl" and nranks > ndev:
        raise SystemExit(f"bench: --ranks threads needs one GPU per rank for RCCL ({nranks} ranks, {ndev} devices visible)")
    result = {}

    def body(ctx, grid, comm):
        out = run_rank(args, comm, ctx, grid, "threads of one process")
        if comm.rank == 0:
            result["out"] = out

    try:
        run_ranks(nprow, npcol, body, device=lambda r: r % ndev, transport=transport)
    except AssertionError as e:
        cause = e.__cause__
        if isinstance(cause, TransportTooSlow):
            print(f"bench: {cause}", file=sys.stderr, flush=True)
            ctypes.CDLL(None).fflush(None)
            os._exit(EXIT_TRANSPORT_TOO_SLOW)
        raise
    ctypes.CDLL(None).fflush(None)
    return result.get("out")
