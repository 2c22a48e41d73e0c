"""First-contact self-tuning of the panel pipeline (bench.py --gpus N, N >= 2).

Three run-time knobs of the distributed HEMM cannot be tuned without the hardware the job runs on - no session of this build
has ever had two GPUs -: the column-panel width of the pipeline (how much of a product's all-reduce can hide behind the next
panel's GEMM against how well a panel fills the chip), the K-piece granularity of products that share the chip with a
collective (CHASE_HIP_PANEL_ROUNDS: what the CUs RCCL's kernels take displace), and whether the row and column communicators
get one communication stream each or share one.  So the first thing a multi-GPU run does with them is MEASURE: in the untimed
warm-up, before its first solve, every candidate setting runs the same few full-width filter steps (both directions of the
distributed HEMM, all-reduces included, linalg/internal/nccl/hemm.hpp:25-399 is the reference's counterpart), one factor at a
time; the fastest setting (max over ranks; exposed communication breaks ties) is locked for everything that follows and the
whole table goes into the JSON line (`autotune`).  In-process setters only (chase_hip_solver_set "panel_cols" /
"panel_rounds", chase_hip_grid_set_comm_streams): no restart, no re-exec.  Every rank applies the same candidate at the same
point of its call sequence and all ranks take the decision from the same agreed numbers, so the collectives stay matched.

Why not "one setting per warm-up ITERATION of a solve": the iterations of a solve are not equal work (widths shrink as columns
retire, iteration 0 carries Lanczos and the first QR), so their seconds would compare settings on different workloads; the
trial steps here are identical for every candidate.  The knobs can still be switched between the iterations of a solve
(tests/dist_scenarios.py scenario_knob_switching)."""

PANELS = (128, 256, 512)
TIE = 0.01                      # relative margin inside which two settings count as equally fast
# A solve is not only filter products: per full-width-equivalent product (one HEMM call; ~160 per solve whatever the size:
# 9 iterations of degree ~20-36 on a shrinking block) it issues about 2.8 SMALL synchronous collectives (~450 per solve: Lanczos
# scalars, agreement collectives, packed Gram / projection sums, residual sums).  A setting is judged by seconds per product
# PLUS that many small-collective latencies: over RCCL's socket transport two communication streams make the filter 10 % faster
# and every small collective 19 ms instead of 1 ms slower (profiles/r05_socket_rccl_streams.txt) - judged on the filter
# alone the tuning would lock a setting that nearly doubles the solve (measured: 20.7 s against 12.4 s).
SMALL_PER_PRODUCT = 2.8


def cost(rec):
    return rec["seconds"] + SMALL_PER_PRODUCT * rec.get("small_collective_us", 0.0) * 1e-6


def streams_candidate_enabled():
    """The two-communication-stream candidate launches the row and the column communicator's collectives concurrently from one
    process, so their device execution order may differ between ranks; it has only ever run over RCCL's socket transport (four
    processes on one GPU).  Until it has been seen on xGMI it is OPT-IN (CHASE_HIP_AUTOTUNE_STREAMS=1) - round-5 advisor; a trial
    that does hang ends in the RCCL watchdog's CHASE_HIP_ECOMM after CHASE_HIP_FABRIC_TIMEOUT_S (grid.hip), not in a re-exec."""
    import os
    return os.environ.get("CHASE_HIP_AUTOTUNE_STREAMS", "0") == "1"


def plan(base, budget, streams=None, skipped=None):
    """One-factor-at-a-time schedule as a list of stages; a stage is (knob, [values to try]); the base value of a knob is always
    measured (first trial overall, then implicitly as the incumbent).  budget = number of trials the caller can afford.
    streams: may the schedule try TWO communication streams (None: streams_candidate_enabled()); going from two streams back to
    one is always allowed.  skipped (a list): receives what the schedule left out and why."""
    if streams is None:
        streams = streams_candidate_enabled()
    stages = [("panel_cols", [p for p in PANELS if p != base["panel_cols"]]),
              ("panel_rounds", [0 if base["panel_rounds"] else 4])]
    if base["comm_streams"] == 2:
        stages.append(("comm_streams", [1]))
    elif streams:
        stages.append(("comm_streams", [2]))
    elif skipped is not None:
        skipped.append({"knob": "comm_streams", "value": 2, "why": "opt-in: CHASE_HIP_AUTOTUNE_STREAMS=1 (never run on xGMI)"})
    out, left = [], budget - 1                # one trial is the base setting itself
    for knob, values in stages:
        take = values[:max(left, 0)]
        if take:
            out.append((knob, take))
            left -= len(take)
        if skipped is not None:
            for v in values[len(take):]:
                skipped.append({"knob": knob, "value": v, "why": "trial budget (%d)" % budget})
    return out


def better(a, b):
    """a, b: trial records {"seconds", "exposed_ms"[, "small_collective_us"]}: is a strictly preferable to b?  Cheaper (seconds
    per product + the small collectives that come with it) by more than the tie margin wins; inside the margin the one with
    less exposed communication wins; equal on both, the incumbent (b) stays."""
    ca, cb = cost(a), cost(b)
    if ca < cb * (1.0 - TIE):
        return True
    if ca > cb * (1.0 + TIE):
        return False
    return a["exposed_ms"] < b["exposed_ms"] * (1.0 - TIE) and ca <= cb * (1.0 + TIE)


def tune(base, budget, measure, streams=None, skipped=None):
    """measure(setting) -> {"seconds", "exposed_ms"} (already agreed between the ranks).  Returns (best setting, table)."""
    best = dict(base)
    rec = measure(best)
    table = [dict(setting=dict(best), **rec, kept=True)]
    best_rec = rec
    if budget < 2:
        return best, table
    for knob, values in plan(base, budget, streams, skipped):
        for v in values:
            cand = dict(best)
            cand[knob] = v
            rec = measure(cand)
            keep = better(rec, best_rec)
            table.append(dict(setting=dict(cand), **rec, kept=keep))
            if keep:
                for row in table[:-1]:
                    row["kept"] = False
                best, best_rec = cand, rec
    return best, table


def apply_setting(s, grid, setting):
    """collective: every rank calls it with the same setting at the same point"""
    s.set(panel_cols=setting["panel_cols"], panel_rounds=setting["panel_rounds"])
    grid.set_comm_streams(setting["comm_streams"])


def current_setting(s, grid):
    return {"panel_cols": int(s.get("panel_cols")), "panel_rounds": int(s.get("panel_rounds")),
            "comm_streams": int(grid.comm_streams())}


def first_contact(s, ctx, grid, comm, nevex, budget=5, steps=2, log=None):
    """Runs the schedule on an initialised distributed solver BEFORE its first solve; leaves the best setting applied.
    A trial = `steps` pairs of full-width filter products (column -> row and row -> column, all-reduces included) between two
    (device sync + barrier) brackets; seconds = max over ranks."""
    import time
    import numpy as np
    from .capi import lib, check

    def measure(setting):
        apply_setting(s, grid, setting)
        s.Start()
        s.initVecs(True)
        check(lib.chase_hip_ctx_set_phase(ctx.h, 1), "set_phase")
        # the pseudo-Hermitian Impl filters with HEMM_H2 (two products and their all-reduces per call: one "pair")
        pseudo = getattr(s, "_colmul", 1) == 2
        pair = ((lambda beta: s.HEMM_H2(nevex, 0.01, beta, -0.3, 0)) if pseudo else
                (lambda beta: (s.HEMM(nevex, 0.01, beta, 0), s.HEMM(nevex, 0.01, -0.5, 0))))
        try:
            pair(0.0)                                      # untimed pair: first touch of this decomposition
            check(lib.chase_hip_grid_wait(grid.h), "grid_wait")
            ctx.sync()
            e0, _ = grid.comm_exposed_ms()
            comm.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                pair(-0.5)
            check(lib.chase_hip_grid_wait(grid.h), "grid_wait")
            ctx.sync()
            dt = time.perf_counter() - t0
            e1, _ = grid.comm_exposed_ms()
        finally:
            check(lib.chase_hip_ctx_set_phase(ctx.h, 0), "set_phase")
        # latency of a small synchronous collective, alternating between the two communicators like the solve does
        from .dist import ROW, COL
        groups = [g for g in (COL, ROW) if lib.chase_hip_grid_group_active(grid.h, g)]
        small_us = 0.0
        if groups:
            scal = ctx.empty((64,), np.float64)
            check(lib.chase_hip_memset(ctx.h, scal.ptr, 0, 64 * 8), "memset")
            for g in groups:
                check(lib.chase_hip_grid_allreduce(grid.h, g, scal.ptr, 64, 0), "allreduce")
            ctx.sync()
            comm.barrier()
            t1 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                for g in groups:
                    check(lib.chase_hip_grid_allreduce(grid.h, g, scal.ptr, 64, 0), "allreduce")
            ctx.sync()
            small_us = (time.perf_counter() - t1) / (reps * len(groups)) * 1e6
            scal.free()
        sec, exp, small_us = comm.allreduce_max([dt, e1 - e0, small_us])
        rec = {"seconds": sec / (2 * steps), "exposed_ms": exp / (2 * steps), "small_collective_us": small_us}
        rec["cost_seconds"] = cost(rec)
        if log:
            log(f"autotune: {setting} -> {rec['seconds'] * 1e3:.1f} ms per full-width product, {rec['exposed_ms']:.1f} ms exposed, "
                f"{small_us:.0f} us per small collective -> cost {rec['cost_seconds'] * 1e3:.1f} ms")
        return rec

    base = current_setting(s, grid)
    skipped = []
    best, table = tune(base, budget, measure, skipped=skipped)
    apply_setting(s, grid, best)
    return {"base": base, "chosen": best, "skipped": skipped,
            "unit": "cost = seconds per full-width distributed HEMM + %.1f x the latency of a small synchronous collective (max over "
                    "ranks); one factor at a time" % SMALL_PER_PRODUCT,
            "trials": table}
