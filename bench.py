#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native ChASE hot path (contract: see the task prompt / DESIGN.md §6).

A "step" is one complete ChASE solve (Lanczos bounds + filter / QR / Rayleigh-Ritz / residual iterations until nev
pairs are locked) of the synthetic workload; the matrix is generated in HBM before the timed region.
  metric  = Chebyshev-filter HEMM GFLOP/s  = 2*F*N^2*(filtered vectors) / time between FilterPhaseStart/End
            (the reference's own model, algorithm/performance.hpp:248-260; F = 4 complex, 1 real)
  extra   = eigenpairs_per_sec (nev / wall per solve), pct of the fp64 MFMA peak, per-phase seconds
N = 1 workload: BASELINE.json configs[1]  (N = 16384 complex Hermitian fp64, nev = 512, nex = 128, one MI355X).
N > 1: 2D block grid of the reference (grid/mpiGrid2D.hpp), one process per GPU, RCCL row/column all-reduces.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X fp64 matrix == vector peak (vendor figure)

# Workload matrix: the Clement-type matrix of the reference's solve tests with its dense Hermitian
# N(0,1)*1e-6 perturbation, scaled by 100/N so that the spectrum spans [-100, 100] like the reference driver's
# --isMatGen matrix (dmax = 100).  Unscaled, ||H|| = N puts the reference's ABSOLUTE tol = 1e-10 below the fp64 residual
# floor at N = 16384 and both the oracle and this backend run into maxIter (measured).
MATRIX_SCALE = 100.0
MATRIX_PERTURB = 1e-6

DEFAULT_WORKLOAD = "cfg2"
# BASELINE.json assigns one configuration to each GPU count (configs[1] 1 GPU, configs[2] 4 GPUs, configs[3] 8 GPUs).
# The default series keeps ONE arithmetic type (complex Hermitian, like the 1- and 8-GPU configurations) so that the
# per-N values are comparable: 2 and 4 GPUs run the complex twin of configs[2] (same N, nev, nex; "cfg3c") on 2x1 / 2x2
# grids; the real-symmetric configs[2] itself is --workload cfg3.  --workload overrides (e.g. cfg4 on every N for a
# strong-scaling series).
DEFAULT_BY_GPUS = {1: "cfg2", 2: "cfg3c", 4: "cfg3c", 8: "cfg4"}
DEFAULT_BLOCK_CYCLIC = {"cfg4": 64}          # BASELINE configs[3]: block-cyclic distribution (nb = 64, examples/1_hello_world)

WORKLOADS = {
    # name: (N, complex, nev, nex)
    "cfg1": (4096, False, 100, 40),
    "cfg2": (16384, True, 512, 128),
    "cfg3": (32768, False, 1024, 256),
    "cfg3c": (32768, True, 1024, 256),
    "cfg4": (65536, True, 2048, 512),
    # BASELINE configs[4]: pseudo-Hermitian Bethe-Salpeter, chase::Solve_pseudo, nex = nev/4 like the other configurations;
    # synthetic matrix chase_hip_gen_bse (dmin 1, dmax 11, off-diagonal 1e-3 N(0,1)); always runs the grid Impl
    "cfg5": (32768, True, 256, 64),
}
PSEUDO_WORKLOADS = {"cfg5"}
BSE_MATRIX = {"dmin": 1.0, "dmax": 11.0, "offdiag": 1e-3}


def mfma_executed_fraction(cplx, m_loc, k_loc):
    """Share of the algorithmic (reference flop model, 4 real multiplications per complex one) flops the filter HEMM
    actually executes on the matrix cores: 3/4 when the three-multiplication complex scheme applies."""
    from chase_amd.capi import lib
    if cplx and lib.chase_hip_gemm3m_enabled() and m_loc % 128 == 0 and k_loc % 8 == 0:
        return 0.75
    return 1.0


def spectrum_check(lam, N, nev):
    """Full-size parity property: the unperturbed Clement-type matrix has the exact spectrum {-N, -N+2, ..., N}
    (SURVEY.md §8c), the 1e-6 dense Hermitian perturbation moves an eigenvalue by O(1e-6) (first order: eps * v^H G v), and
    the bench matrix is that matrix times MATRIX_SCALE / N.  Returns the largest deviation of the nev computed
    eigenvalues from the analytic ones and the bound it must meet."""
    scale = MATRIX_SCALE / N
    exact = scale * (-N + 2.0 * np.arange(nev))
    dev = float(np.max(np.abs(np.sort(np.asarray(lam)[:nev]) - exact)))
    tol = 50.0 * MATRIX_PERTURB * scale
    return {"max_abs_dev_from_analytic": dev, "bound": tol, "ok": bool(dev <= tol)}


def cpu_baseline(N, cplx, ncols, budget_s=25.0):
    """Times the CPU oracle's filter HEMM (oracle/chase_oracle.py: OracleCPU.HEMM -> numpy/OpenBLAS gemm) on a bounded
    sample of the same workload: full-height H, as many columns as fit the time budget."""
    from oracle import chase_oracle as O
    threads = os.cpu_count() or 1
    F = 4 if cplx else 1
    # bound the sample: H is N x N (the oracle needs it on the host); shrink N if the host cannot hold it comfortably
    n_s = N
    while n_s * n_s * (16 if cplx else 8) > 6e9:
        n_s //= 2
    cols = min(ncols, 128)
    H = O.clement(n_s, cplx, perturb=0)
    k = O.OracleCPU(H, cols // 2, cols - cols // 2)
    rng = np.random.default_rng(0)
    k.V1[:] = rng.standard_normal(k.V1.shape)
    k.V2[:] = k.V1
    t0 = time.perf_counter()
    k.HEMM(cols, 0.01, 0.0, 0)                       # warm-up / thread pool start
    t_first = time.perf_counter() - t0
    reps = int(max(1, min(20, (budget_s - t_first) / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        k.HEMM(cols, 0.01, -0.5, 0)
    dt = (time.perf_counter() - t0) / reps
    gflops = 2.0 * F * n_s * n_s * cols / dt / 1e9
    out = {"value": gflops, "unit": "GFLOP/s", "cores": threads, "kind": "port",
           "sample": f"oracle HEMM (numpy/OpenBLAS zgemm)" if cplx else "oracle HEMM (numpy/OpenBLAS dgemm)",
           "sample_shape": {"N": n_s, "ncols": cols, "reps": reps, "seconds_per_call": dt}}
    # second bounded sample: one complete oracle solve of the reference's CPU-runnable configuration (BASELINE configs[0]:
    # N = 4096 real, nev = 100, nex = 40; the reference itself measured 5.44 s on 8 vCPU)
    try:
        t0 = time.perf_counter()
        Hs = O.clement(4096, False, perturb=0)
        ks = O.OracleCPU(Hs, 100, 40)
        so = O.solve(ks)
        ts = time.perf_counter() - t0
        out["solve_sample"] = {"workload": "cfg1 (N=4096 real, nev=100, nex=40), oracle solve", "seconds": ts,
                               "eigenpairs_per_sec": 100.0 / ts, "iterations": so["iterations"],
                               "filtered_vecs": so["filtered_vecs"]}
    except Exception as e:  # the HEMM sample above is the contract; this one is informative
        out["solve_sample"] = {"error": str(e)}
    return out


def run_single(args):
    from chase_amd.capi import Context, Solver, lib, check
    N, cplx, nev, nex = WORKLOADS[args.workload]
    if args.n:
        N = args.n
    dt = np.complex128 if cplx else np.float64
    ctx = Context(0)
    info = ctx.info()
    dH = ctx.gen_clement(N, cplx, scale=MATRIX_SCALE / N, perturb=MATRIX_PERTURB, seed=42)
    ctx.sync()
    s = Solver(ctx, None, nev, nex, h_on_device_ptr=dH.ptr, N=N, cplx=cplx)
    s.set(device_rng=1)
    F = 4 if cplx else 1
    stats = []
    for it in range(args.warmup):
        s.set(reset_counters=1)
        s.solve()
    ctx.sync()
    t0 = time.perf_counter()
    for it in range(args.steps):
        s.set(reset_counters=1)
        st = s.solve()
        st["hemm_calls"] = s.get("hemm_calls")
        st["hemm_reused_vecs"] = s.get("hemm_reused_vecs")
        stats.append(st)
    ctx.sync()
    wall = time.perf_counter() - t0
    # vectors that went through a filter HEMM: the reference's count minus the first-step columns served from the
    # Rayleigh-Ritz product - those cost O(N n), crediting them N^2 flops would inflate the rate
    reused = sum(x["hemm_reused_vecs"] for x in stats)
    vecs = sum(x["filtered_vecs"] for x in stats) - reused
    filt_s = sum(x["filter_ms_device"] for x in stats) * 1e-3
    calls = sum(x["hemm_calls"] for x in stats)
    flops = 2.0 * F * N * N * vecs
    gflops = flops / filt_s / 1e9
    xf = mfma_executed_fraction(cplx, N, N)
    # parity guard inside the bench: the timed solves must have converged to the solver tolerance
    resid = s.resid()[:nev]
    lam = s.ritzv[:nev].copy()
    spec = spectrum_check(lam, N, nev)
    ok = bool(np.all(np.isfinite(lam)) and np.max(resid) < 1e-8 and stats[-1]["locked"] >= nev and spec["ok"])
    last = stats[-1]
    out = {
        "metric": "chebyshev_filter_hemm_gflops", "value": gflops, "unit": "GFLOP/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if args.user_workload else "weak", "vs_baseline": None,
        "dtype": "complex f64" if cplx else "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: ChASE solve, perturbed Clement-type Hermitian (x100/N) N={N} "
                               f"{'complex' if cplx else 'real'} fp64, nev={nev} nex={nex}, tol 1e-10, deg 20 opt, 1x1 grid",
                   "N": N, "nev": nev, "nex": nex, "grid": "1x1"},
        "eigenpairs_per_sec": nev / (wall / args.steps),
        "pct_fp64_mfma_peak": 100.0 * xf * gflops / 1e3 / FP64_MFMA_PEAK_TFLOPS,
        "mfma_executed_fraction": xf,
        "converged": ok, "max_resid": float(np.max(resid)), "spectrum_check": spec,
        "iterations": last["iterations"], "filtered_vecs_per_solve": (vecs + reused) / args.steps,
        "hemm_vecs_per_solve": vecs / args.steps, "first_step_vecs_from_rr_per_solve": reused / args.steps,
        "phase_seconds_last_solve": {k: last[k] for k in ("t_all", "t_init", "t_lanczos", "t_filter", "t_qr", "t_rr", "t_resid")},
        "device": info["name"],
        "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel<cplx,op=N,TAG=1> (filter HEMM)",
                     "achieved": gflops / 1e3, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": gflops / 1e3 / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                     "executed": xf * gflops / 1e3, "executed_frac": xf * gflops / 1e3 / FP64_MFMA_PEAK_TFLOPS,
                     "note": "achieved = ALGORITHMIC flops (reference model 2*F*N^2*ncols, F = 4 complex) / time; the "
                             "complex filter kernel forms each complex product from 3 real MFMA products (3M), so the "
                             "matrix cores execute `executed` = 3/4 of that; executed_frac is the MFMA utilisation",
                     "launches": calls, "avg_launch_ms": filt_s * 1e3 / max(calls, 1),
                     "launch_unit": "one HEMM call = whole-tile kernel (+ ragged-column kernel when the width is not a "
                                    "multiple of the tile width) + tail reduce; rocprofv3: sum over the TAG=1 kernels",
                     "flop_per_launch_avg": flops / max(calls, 1)},
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get("workload") == args.workload:
                out["roofline"]["traffic"] = rec.get("hbm_bytes_per_launch")
        except Exception:
            pass
    s.close()
    ctx.close()
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, cplx, nev + nex, args.cpu_budget)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--n", "--size", dest="n", type=int, default=0, help="override N (development only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=25.0)
    ap.add_argument("--dist", action="store_true",
                    help="run the grid Impl (pChaseHip) even on one GPU (1x1 grid; development: panel-pipeline overheads)")
    ap.add_argument("--block-cyclic", type=int, default=-1,
                    help="block size of a block-cyclic H distribution (0 = block layout, -1 = the workload's default)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (args.workload in PSEUDO_WORKLOADS or args.dist) and world == 1 and args.gpus <= 1:
        # the pseudo-Hermitian workload runs the grid Impl on a 1x1 grid (communicator-free) when launched directly
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0")
        os.environ["WORLD_SIZE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port))
        args.gpus = 1
        world = 1
        from chase_amd.dist_bench import run_distributed
        print(json.dumps(run_distributed(args)), flush=True)
        return
    if args.gpus > 1 or world > 1:
        from chase_amd.dist_bench import run_distributed
        out = run_distributed(args)
        if out is not None:
            print(json.dumps(out), flush=True)
        return
    args.user_workload = args.workload is not None
    if args.workload is None:
        args.workload = DEFAULT_WORKLOAD
    out = run_single(args)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
