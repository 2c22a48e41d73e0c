"""BASELINE.json's multi-GPU configurations AT FULL SIZE on one GPU: the ranks of the grid are threads of this process
(tests/rank_threads.py, host-callback transport), every rank holds its shard of the bench matrix in HBM, and the solve is the
one `bench.py --gpus N` runs (same generators, same settings).  What the reference's distributed solve tests assert
(tests/chase_distributed_solve.cpp:209-284: every rank recomputes H v - lambda v from the solved block and checks it, the
eigenvalues against the known spectrum; tests/chase_distributed_solve_pseudo_bse_test.cpp: residuals + positive spectrum)
is asserted here at the sizes BASELINE.json names instead of N = 1001.

Used by tests/test_gpu_fullsize.py (driver-run) and scripts/dev_rehearsal_threads.py (prints the record)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def bse_diagonal(N, nev, dmin, dmax):
    """The nev smallest diagonal entries of the A block of chase_hip_gen_bse (gen_kernels.hip: sqrt(dmin^2 + (dmax^2 - dmin^2)
    i / (h - 1)), h = N / 2): the positive spectrum of the unperturbed matrix."""
    h = N // 2
    i = np.arange(nev, dtype=np.float64)
    return np.sqrt(dmin * dmin + (dmax * dmax - dmin * dmin) * i / (h - 1))


def fullsize_rank(ctx, grid, comm, wl, nb, result, hash_replicas=True, perturb=None, device_rng=1):
    """One rank of the full-size solve; rank 0 leaves the record in `result` (a dict shared by the rank threads).
    perturb / device_rng: the oracle-pinned variant runs the UNPERTURBED matrix from the reference's start vectors
    (mt19937(1337 + grid row) on the host, pchase_cpu.hpp:272-283) so that an independent implementation can be compared."""
    import bench as B
    from chase_amd import dist as cd
    N, cplx, nev, nex = B.WORKLOADS[wl]
    nprow, npcol = grid.nprow, grid.npcol
    rl, cl = cd.Layout(N, nb, nprow), cd.Layout(N, nb, npcol)
    pseudo = wl in B.PSEUDO_WORKLOADS
    if pseudo:
        dH = cd.gen_bse_local(ctx, N, cplx, rl, cl, grid.myrow, grid.mycol, **B.BSE_MATRIX)
        ctx.sync()
        s = cd.DistPseudoSolver(ctx, grid, dH, N, nev, nex, cplx, nb, nb)
        s.set(device_rng=1, numlanczos=10, lanczositer=50)      # the reference's BSE settings (5_bse_benchmark / BSE test)
    else:
        dH = cd.gen_clement_local(ctx, N, cplx, rl, cl, grid.myrow, grid.mycol, scale=B.MATRIX_SCALE / N,
                                  perturb=B.MATRIX_PERTURB if perturb is None else perturb)
        ctx.sync()
        s = cd.DistSolver(ctx, grid, dH, N, nev, nex, cplx, nb, nb)
        s.set(device_rng=device_rng)
    comm.barrier()
    t = time.perf_counter()
    st = s.solve()
    ctx.sync()
    comm.barrier()
    wall = time.perf_counter() - t
    lam = s.ritzv[:nev].copy()
    resid = s.resid()[:nev].copy()
    # replicas: the column-type eigenvector block of a grid row is held by every member of that row group
    digest = None
    if hash_replicas:
        # on the device, 8 bytes back per rank (round 4 downloaded 8 x 0.67 GB and hashed them on the host)
        digest = s.hash_V(nev)
    resid_re = s.recompute_residuals(nev, lam)                  # fresh four-product H V (collective)
    everyone = comm.all_gather_object((grid.myrow, grid.mycol, lam, resid, digest, resid_re))
    if comm.rank == 0:
        tol = s.get("tol")
        worst_re = np.max(np.stack([e[5] for e in everyone]), axis=0)
        # pairs the solver took as converged (residual <= tol; stagnating pairs are locked above it on purpose,
        # algorithm.inc:519-578) whose independently recomputed residual is above tol: two correct four-product
        # evaluations of one residual differ by ~1e-14 ||H||, i.e. ~1e-4 tol, so the bar is tol (1 + 1e-3)
        above = int(np.sum((resid <= tol) & (worst_re > tol * (1 + 1e-3))))
        lam_equal = all(np.array_equal(everyone[0][2], e[2]) for e in everyone)
        resid_equal = all(np.array_equal(everyone[0][3], e[3]) for e in everyone)
        rows = {}
        for (i, j, _, _, dg, _) in everyone:
            rows.setdefault(i, set()).add(dg)
        result.update(workload=wl, grid=f"{nprow}x{npcol}", nb=nb, N=N, nev=nev, nex=nex,
                      transport=("device-side collectives between the rank threads' buffers (shared-device transport), ONE GPU"
                                 if grid.transport == "shared" else "host callbacks, ranks = threads of one process, ONE GPU"),
                      iterations=st["iterations"], filtered_vecs=st["filtered_vecs"], locked=st["locked"],
                      wall_seconds=wall, max_resid=float(np.max(resid)),
                      max_resid_recomputed=float(np.max(worst_re)), tol=tol,
                      pairs_converged_by_solver_but_recomputed_above_tol=above,
                      residuals_rechecked_on_the_tolerance=int(s.get("resd_rechecked")),
                      eigenvalues_bitwise_equal_on_all_ranks=bool(lam_equal and resid_equal),
                      eigenvector_replicas_bitwise_equal=(all(len(v) == 1 for v in rows.values()) if hash_replicas else None),
                      spectrum_check=None if pseudo else B.spectrum_check(lam, N, nev),
                      max_abs_dev_from_analytic=(None if pseudo else
                                                 float(np.max(np.abs(np.sort(lam) - (B.MATRIX_SCALE / N) * (-N + 2.0 * np.arange(nev)))))),
                      lambda_first=lam[:4].tolist(), lambda_last=lam[-2:].tolist(),
                      ascending=bool(np.all(np.diff(lam) >= 0)),
                      phases={k: st[k] for k in B.PHASES})
        if pseudo:
            d = bse_diagonal(N, nev, B.BSE_MATRIX["dmin"], B.BSE_MATRIX["dmax"])
            result["bse_max_dev_from_unperturbed_diagonal"] = float(np.max(np.abs(lam - d)))
    s.close()
    del dH


def pseudo_oracle_rank(ctx, grid, comm, N, nev, nex, bse, result):
    """One rank of the ORACLE-PINNED full-size pseudo-Hermitian solve: the shard of oracle.synthetic_bse_block (a matrix every rank
    can build for itself, and the CPU oracle as a whole), block layout, the reference's start vectors (host mt19937(1337 + grid
    row), global lower half damped: pchase_cpu.hpp:272-311), the reference's BSE settings."""
    from chase_amd import dist as cd
    from oracle import chase_oracle as O
    rl, cl = cd.Layout(N, 0, grid.nprow), cd.Layout(N, 0, grid.npcol)
    rows, cols = rl.globals_of(grid.myrow), cl.globals_of(grid.mycol)
    blk = np.empty((len(rows), len(cols)), dtype=np.complex128, order="F")
    for r0 in range(0, len(rows), 512):                        # in row chunks: the index / hash temporaries are chunk-sized
        blk[r0:r0 + 512, :] = O.synthetic_bse_block(rows[r0:r0 + 512], cols, N, **bse)
    dH = ctx.array(blk)
    del blk
    s = cd.DistPseudoSolver(ctx, grid, dH, N, nev, nex, True, 0, 0)
    s.set(numlanczos=10, lanczositer=50)                        # host RNG: device_rng stays off
    comm.barrier()
    t = time.perf_counter()
    st = s.solve()
    ctx.sync()
    comm.barrier()
    wall = time.perf_counter() - t
    lam, resid = s.ritzv[:nev].copy(), s.resid()[:nev].copy()
    digest = s.hash_V(nev)
    resid_re = s.recompute_residuals(nev, lam)
    everyone = comm.all_gather_object((grid.myrow, grid.mycol, lam, digest, resid_re))
    if comm.rank == 0:
        rows_d = {}
        for (i, j, _, dg, _) in everyone:
            rows_d.setdefault(i, set()).add(dg)
        result.update(N=N, nev=nev, nex=nex, grid=f"{grid.nprow}x{grid.npcol}", iterations=st["iterations"],
                      filtered_vecs=st["filtered_vecs"], locked=st["locked"], wall_seconds=wall, max_resid=float(np.max(resid)),
                      max_resid_recomputed=float(np.max(np.stack([e[4] for e in everyone]))),
                      lambda_first=lam[:6].tolist(), lambda_last=lam[-2:].tolist(), lambda_sum=float(np.sum(lam)),
                      eigenvalues_bitwise_equal_on_all_ranks=all(np.array_equal(everyone[0][2], e[2]) for e in everyone),
                      eigenvector_replicas_bitwise_equal=all(len(v) == 1 for v in rows_d.values()),
                      phases={k: st[k] for k in ("t_all", "t_lanczos", "t_filter", "t_qr", "t_rr", "t_resid")})
    s.close()
    dH.free()


def run_fullsize(wl, nprow, npcol, nb, hash_replicas=True, transport="shared", **kw):
    """transport "shared" (round 5): the rank threads' collectives are device-side sums / copies ordered by events between their
    streams (chase_hip_grid_create_shared) - asynchronous like RCCL; "host": staged through pinned host memory into the Python
    fabric (rounds 3-4; a fifth of the full-size wall time was that staging)"""
    from rank_threads import run_ranks
    result = {}
    run_ranks(nprow, npcol, fullsize_rank, wl, nb, result, hash_replicas, transport=transport, **kw)
    return result
