"""Full-size dress rehearsal of a multi-GPU configuration on ONE GPU in ONE process: the ranks of the grid are threads (tests/
rank_threads.py, host-callback transport), each holds its shard of the bench matrix.  Default: BASELINE configs[3] exactly as the
8-GPU job runs it (N = 65536 complex, nev = 2048, nex = 512, 4 x 2 grid, block-cyclic nb = 64): 8 x 17 GB in the 288 GB.
usage: dev_rehearsal_threads.py [workload] [nprow npcol] [nb]"""
import json
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B
from chase_amd import dist as cd
from rank_threads import run_ranks

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
nprow, npcol = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 2)
nb = int(sys.argv[4]) if len(sys.argv) > 4 else B.DEFAULT_BLOCK_CYCLIC.get(wl, 0)
N, cplx, nev, nex = B.WORKLOADS[wl]
result = {}


def rank_main(ctx, grid, comm):
    rl, cl = cd.Layout(N, nb, nprow), cd.Layout(N, nb, npcol)
    pseudo = wl in B.PSEUDO_WORKLOADS
    if pseudo:
        dH = cd.gen_bse_local(ctx, N, cplx, rl, cl, grid.myrow, grid.mycol, **B.BSE_MATRIX)
        ctx.sync()
        s = cd.DistPseudoSolver(ctx, grid, dH, N, nev, nex, cplx, nb, nb)
        s.set(device_rng=1, numlanczos=10, lanczositer=50)
    else:
        dH = cd.gen_clement_local(ctx, N, cplx, rl, cl, grid.myrow, grid.mycol, scale=B.MATRIX_SCALE / N, perturb=B.MATRIX_PERTURB)
        ctx.sync()
        s = cd.DistSolver(ctx, grid, dH, N, nev, nex, cplx, nb, nb)
        s.set(device_rng=1)
    comm.barrier()
    t = time.perf_counter()
    st = s.solve()
    ctx.sync(); comm.barrier()
    wall = time.perf_counter() - t
    lam = s.ritzv[:nev].copy()
    resid_re = s.recompute_residuals(nev, lam)
    lams = comm.all_gather_object(lam)
    assert all(np.array_equal(lams[0], l) for l in lams), "ranks disagree on the eigenvalues"
    if comm.rank == 0:
        result.update(workload=wl, grid=f"{nprow}x{npcol}", nb=nb, N=N, nev=nev, nex=nex, transport="host callbacks, ranks = threads of one process, ONE GPU",
                      iterations=st["iterations"], filtered_vecs=st["filtered_vecs"], locked=st["locked"], wall_seconds=wall,
                      max_resid=float(np.max(s.resid()[:nev])), max_resid_recomputed=float(np.max(resid_re)),
                      spectrum_check=None if pseudo else B.spectrum_check(lam, N, nev), lambda_first=lam[:4].tolist(),
                      phases={k: st[k] for k in B.PHASES})
    s.close()


run_ranks(nprow, npcol, rank_main)
print(json.dumps(result), flush=True)
