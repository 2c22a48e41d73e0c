"""Single-rank replay of a multi-GPU solve on ONE GPU (`bench.py --replay-rank 4x2`).

What it measures: everything ONE rank of the nprow x npcol grid executes during a solve of the workload - its local H block
(16384 x 32768 complex at config 4 on 4 x 2, block-cyclic nb = 64), its column-type / row-type vector blocks, the panel
products of the pipelined HEMM (linalg/internal/nccl/hemm.hpp:25-399 is what such a rank runs in the reference), CholQR on its
rows, Rayleigh-Ritz with the replicated projected eigensolver, residuals, Lanczos (Impl/pchase_gpu/pchase_gpu.hpp:1550-1700) -
with nothing else on the GPU and NO communication: the grid is a loopback grid (chase_hip_grid_create_loopback: collectives keep
their streams, events and waits but move nothing).  The call sequence is the one of a REAL solve of the workload: a scalar tape
(chase_amd/host/tape.hpp) recorded on the single-GPU solve is replayed, so the unmodified driver issues exactly the recorded
virtual calls (same degrees, same locking, same widths) whatever the lone rank computes.  Numbers are wrong by construction;
launches, shapes and time are the real rank's.  T_rank(grid) is therefore the COMPUTE SIDE of the multi-GPU solve:
single-GPU solve seconds / (ranks * T_rank) bounds the parallel efficiency from above, and what xGMI adds is exposed waits.
"""
import json
import os
import sys
import time

import numpy as np


def record_tape(ctx, workload, n_override=0, log=None):
    """One real single-GPU solve of the workload with the tape recording; returns (tape, meta)."""
    import bench as B
    from .capi import Solver, tape_mode, tape_get
    N, cplx, nev, nex = B.WORKLOADS[workload]
    if n_override:
        N = n_override
    pseudo = workload in B.PSEUDO_WORKLOADS
    grid = None
    if pseudo:
        # the pseudo-Hermitian workload always runs the grid Impl (bench.py): its single-GPU solve is that Impl on a 1 x 1 grid,
        # where a loopback transport is exact (every group has one member)
        from . import dist as cd
        grid = cd.Grid(ctx, 1, 1, 0, transport="loopback")
        lay = cd.Layout(N, 0, 1)
        dH = cd.gen_bse_local(ctx, N, cplx, lay, lay, 0, 0, **B.BSE_MATRIX)
        ctx.sync()
        s = cd.DistPseudoSolver(ctx, grid, dH, N, nev, nex, cplx, 0, 0)
        s.set(device_rng=1, numlanczos=10, lanczositer=50)
    else:
        dH = ctx.gen_clement(N, cplx, scale=B.MATRIX_SCALE / N, perturb=B.MATRIX_PERTURB, seed=42)
        ctx.sync()
        s = Solver(ctx, None, nev, nex, h_on_device_ptr=dH.ptr, N=N, cplx=cplx)
        s.set(device_rng=1)
    tape_mode(s, 1)
    if log:
        s.set_iteration_hook(lambda it, f, l, u: log(f"record: iteration {it}: {f} vectors filtered, {l} locked") or False)
    t0 = time.perf_counter()
    st = s.solve()
    ctx.sync()
    wall = time.perf_counter() - t0
    tape = tape_get(s)
    lam, resid = s.ritzv[:nev].copy(), s.resid()[:nev].copy()
    spec = None if pseudo else B.spectrum_check(lam, N, nev)
    meta = {"workload": workload, "pseudo": pseudo, "N": N, "cplx": bool(cplx), "nev": nev, "nex": nex,
            "iterations": st["iterations"], "filtered_vecs": st["filtered_vecs"], "locked": st["locked"],
            "solve_seconds": st["t_all"], "wall_seconds": wall, "phases": {k: st[k] for k in B.PHASES},
            "filter_seconds_device": st["filter_ms_device"] * 1e-3,
            "max_resid": float(np.max(resid)), "spectrum_check": spec, "device": ctx.info()["name"],
            "residuals_rechecked_on_the_tolerance": int(s.get("resd_rechecked"))}
    s.close()
    dH.free()
    if grid is not None:
        grid.close()
    return tape, meta


def save_tape(path, tape, meta):
    np.savez_compressed(path, tape=np.asarray(tape, dtype=np.float64), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


def load_tape(path):
    z = np.load(path)
    return z["tape"], json.loads(bytes(z["meta"]).decode())


def replay_rank(ctx, tape, meta, nprow, npcol, rank=0, block_cyclic=None, oplog=False, log=None, settings=None):
    """Drives rank `rank` of the nprow x npcol grid through the taped solve on a loopback grid; returns the record."""
    import bench as B
    from . import dist as cd
    from .capi import tape_mode, tape_load, gemm_counters
    workload, N, cplx, nev, nex = meta["workload"], meta["N"], meta["cplx"], meta["nev"], meta["nex"]
    nb = B.DEFAULT_BLOCK_CYCLIC.get(workload, 0) if block_cyclic is None else block_cyclic
    grid = cd.Grid(ctx, nprow, npcol, rank, transport="loopback")
    grid.set_profiling(True)
    if settings and settings.get("comm_streams"):
        grid.set_comm_streams(settings["comm_streams"])
    model = (settings or {}).get("loopback_model")
    if model:
        grid.set_loopback_model(model.get("busbw_GBps", 0.0), model.get("latency_us", 0.0), model.get("touch", False),
                                model.get("workgroups", 0))
    rl, cl = cd.Layout(N, nb, nprow), cd.Layout(N, nb, npcol)
    if meta.get("pseudo"):
        dH = cd.gen_bse_local(ctx, N, cplx, rl, cl, grid.myrow, grid.mycol, **B.BSE_MATRIX)
        ctx.sync()
        s = cd.DistPseudoSolver(ctx, grid, dH, N, nev, nex, cplx, nb, nb)
        s.set(device_rng=1, numlanczos=10, lanczositer=50)
    else:
        dH = cd.gen_clement_local(ctx, N, cplx, rl, cl, grid.myrow, grid.mycol, scale=B.MATRIX_SCALE / N, perturb=B.MATRIX_PERTURB)
        ctx.sync()
        s = cd.DistSolver(ctx, grid, dH, N, nev, nex, cplx, nb, nb)
        s.set(device_rng=1)
    if settings and "pipeline" in settings:
        s.set(pipeline=settings["pipeline"])
    if settings and settings.get("panel_cols"):
        s.set(panel_cols=settings["panel_cols"])
    if settings and settings.get("panel_rounds") is not None:
        s.set(panel_rounds=settings["panel_rounds"])
    tuned = None
    if settings and settings.get("autotune"):
        # the first-contact self-tuning of the multi-GPU bench (chase_amd/autotune.py) on this lone rank: its trial steps run
        # against the MODELLED collectives, which exercises the whole mechanism (in-process setters, identical trial steps,
        # selection) on one GPU; on the real node the same code runs on every rank with max-over-ranks timings
        from .autotune import first_contact

        class _Solo:
            rank, world = 0, 1
            def barrier(self): pass
            def allreduce_max(self, v): return [float(x) for x in v]

        tuned = first_contact(s, ctx, grid, _Solo(), nev + nex, budget=int(settings["autotune"]), log=log)
        s.set(reset_counters=1)
    tape_load(s, tape)
    tape_mode(s, 2)
    per_iter = []
    last = [time.perf_counter()]

    def hook(it, filtered, locked, unconverged):
        now = time.perf_counter()
        per_iter.append({"iteration": it, "filtered_vecs": filtered, "locked": locked, "seconds": now - last[0]})
        last[0] = now
        if log:
            log(f"replay {nprow}x{npcol}: iteration {it}: {filtered} vectors filtered, {locked} locked, {now - t0:.1f} s")
        return False

    s.set_iteration_hook(hook)
    for ph in range(4):
        gemm_counters(ctx, ph, reset=True)
    if oplog:
        ctx.oplog(True)
    ctx.sync()
    t0 = last[0] = time.perf_counter()
    st = s.solve()
    ctx.sync()
    wall = time.perf_counter() - t0
    lines = None
    if oplog:
        ctx.oplog(False)
        lines = ctx.oplog_lines()
    exposed_ms, waits = grid.comm_exposed_ms()
    books = {}
    for ph, name in ((0, "other"), (1, "filter"), (2, "h_times_block_outside_filter"), (3, "verification")):
        m, e, n = gemm_counters(ctx, ph)
        books[name] = {"model_flops": m, "executed_flops": e, "products": n}
    filt_s = st["filter_ms_device"] * 1e-3
    ok = (st["iterations"] == meta["iterations"] and st["filtered_vecs"] == meta["filtered_vecs"]
          and st["locked"] == meta["locked"] and int(s.get("tape_position")) == int(s.get("tape_size")))
    rec = {"grid": f"{nprow}x{npcol}", "rank": rank, "coords": [grid.myrow, grid.mycol], "block_cyclic_nb": nb,
           "local_shape_H": [s.m_loc, s.n_loc], "H_loc_GB": s.m_loc * s.n_loc * (16 if cplx else 8) / 1e9,
           "T_rank_seconds": st["t_all"], "wall_seconds": wall, "phases": {k: st[k] for k in B.PHASES},
           "filter_seconds_device": filt_s,
           "filter_tflops_model_this_rank": books["filter"]["model_flops"] / filt_s / 1e12 if filt_s > 0 else None,
           "filter_tflops_executed_this_rank": books["filter"]["executed_flops"] / filt_s / 1e12 if filt_s > 0 else None,
           "iterations": st["iterations"], "filtered_vecs": st["filtered_vecs"], "locked": st["locked"],
           "call_sequence_equals_recording": bool(ok), "qr_variant_mismatches": int(s.get("tape_qr_mismatches")),
           "qr_shifted_refactorisations_on_replayed_numbers": int(s.get("tape_qr_retries")),
           "projected_matrices_replaced_by_the_identity": int(s.get("tape_tolerated")),
           "residuals_rechecked": int(s.get("resd_rechecked")),
           "waits_on_communication_streams": int(waits), "exposed_ms_of_those_waits_with_nothing_on_the_wire": exposed_ms,
           "gemm_books": books, "per_iteration": per_iter,
           "comm_streams": grid.comm_streams(),
           "collectives": ("none enqueued (compute side alone)" if not model else
                           "MODELLED: each holds its stream and %d workgroups (512 threads, 128 VGPRs) for %.0f us + wire bytes / "
                           "%.0f GB/s bus bandwidth%s"
                           % (model.get("workgroups", 0) or 32, model.get("latency_us", 0.0), model.get("busbw_GBps", 0.0),
                              ", plus one read+write pass over the payload" if model.get("touch") else "")),
           "autotune": tuned, "loopback_model": model, "settings": {k: v for k, v in (settings or {}).items() if k != "loopback_model"}}
    if lines is not None:
        rec["oplog_lines"] = len(lines)
    s.close()
    dH.free()
    grid.close()
    return rec, lines


def run(args):
    """bench.py --replay-rank GRIDS [--tape FILE] [--replay-workload W]: record (or load) the tape, replay one rank of each
    grid, print ONE JSON line."""
    import bench as B
    from .capi import Context

    def log(msg):
        print("bench: " + msg, file=sys.stderr, flush=True)

    workload = args.workload or B.DEFAULT_WORKLOAD
    ctx = Context(0)
    tape = meta = None
    if args.tape and os.path.exists(args.tape):
        tape, meta = load_tape(args.tape)
        if meta["workload"] != workload or (args.n and meta["N"] != args.n):
            raise SystemExit(f"bench: tape {args.tape} was recorded for {meta['workload']} N={meta['N']}")
        log(f"tape loaded from {args.tape}: {meta['iterations']} iterations, {meta['filtered_vecs']} vectors, "
            f"single-GPU solve {meta['solve_seconds']:.1f} s")
    else:
        log(f"recording the tape: one real single-GPU solve of {workload}")
        tape, meta = record_tape(ctx, workload, args.n, log)
        log(f"recorded: {meta['iterations']} iterations, {meta['filtered_vecs']} vectors, {meta['solve_seconds']:.1f} s, "
            f"{tape.size} doubles")
        if args.tape:
            save_tape(args.tape, tape, meta)
    out = {"metric": "single_rank_replay_seconds", "unit": "s", "data": "synthetic", "n_gpus": 1,
           "config": {"workload": f"{workload}: single-rank replay of the taped solve on a loopback grid (no communication): "
                                  "the compute side of one rank of the multi-GPU solve", "N": meta["N"], "nev": meta["nev"],
                      "nex": meta["nex"]},
           "single_gpu": meta, "replays": []}
    variants = [None]
    if args.loopback_busbw:
        variants = [None if float(b) <= 0 else {"busbw_GBps": float(b), "latency_us": args.loopback_latency_us, "touch": True,
                                                "workgroups": int(w)}
                    for b in args.loopback_busbw.split(",") for w in str(args.loopback_wgs).split(",")]
        variants = [v for i, v in enumerate(variants) if v is not None or None not in variants[:i]]
    base_settings = {}
    if args.replay_panel:
        base_settings["panel_cols"] = args.replay_panel
    if args.replay_comm_streams:
        base_settings["comm_streams"] = args.replay_comm_streams
    if args.replay_no_pipeline:
        base_settings["pipeline"] = 0
    if args.replay_panel_rounds >= 0:
        base_settings["panel_rounds"] = args.replay_panel_rounds
    if args.replay_autotune:
        base_settings["autotune"] = args.replay_autotune
    for spec, model in [(g, m) for g in args.replay_rank.split(",") for m in variants]:
        r, c = (int(x) for x in spec.lower().split("x"))
        rec, lines = replay_rank(ctx, tape, meta, r, c, rank=args.replay_rank_index, oplog=bool(args.oplog_out), log=log,
                                 settings=dict(base_settings, loopback_model=model))
        ranks = r * c
        rec["compute_side_speedup_bound"] = meta["solve_seconds"] / rec["T_rank_seconds"]
        rec["compute_side_efficiency_bound"] = meta["solve_seconds"] / (ranks * rec["T_rank_seconds"])
        out["replays"].append(rec)
        log(f"replay {spec} [{rec['collectives'][:46]}, {(model or {}).get('busbw_GBps', 0):.0f} GB/s]: exposed {rec['exposed_ms_of_those_waits_with_nothing_on_the_wire']:.0f} ms; T_rank = {rec['T_rank_seconds']:.2f} s (single GPU {meta['solve_seconds']:.1f} s): compute-side "
            f"speed-up bound {rec['compute_side_speedup_bound']:.2f}x of {ranks}, phases {rec['phases']}")
        if args.oplog_out and lines is not None:
            with open(args.oplog_out.replace("%g", spec), "w") as f:
                f.write("\n".join(lines) + "\n")
    out["value"] = out["replays"][-1]["T_rank_seconds"]
    ctx.close()
    return out
