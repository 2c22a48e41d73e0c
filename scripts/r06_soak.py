"""Soak of the plane-fed 3M loop: the same product launched many times (full width and a ragged, K-split shape) must give the same
64-bit content hash every time (a race on the plane ring or on a refilled fragment would show as run-to-run differences), and
repeated config-2 / config-3c solves must be bitwise identical in eigenvalues, residuals, counts and eigenvector hash."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from chase_amd.capi import Context, Solver, lib, check  # noqa: E402
with Context(0) as ctx:
    lib.chase_hip_ctx_set_phase(ctx.h, 1)
    for (op, m, n, k, reps) in (("N", 65536, 2560, 65536, 12), ("N", 16384, 133, 16384, 200), ("C", 8192, 640, 32768, 100), ("N", 16384, 256, 32768, 100)):
        ra = (m, k) if op == "N" else (k, m)
        dA = ctx.empty(ra, np.complex128); dB = ctx.empty((k, n), np.complex128); dC = ctx.empty((m, n), np.complex128)
        check(lib.chase_hip_fill_normal(ctx.h, 1, ra[0], ra[1], dA.ptr, ra[0], 0, 0, ra[0], 1), "fill")
        check(lib.chase_hip_fill_normal(ctx.h, 1, k, n, dB.ptr, k, 0, 0, k, 2), "fill")
        if n == 256:
            lib.chase_hip_ctx_set_gemm_min_rounds(ctx.h, 4)
        hashes = set()
        for r in range(reps):
            ctx.gemm(op, m, n, k, 0.5, dA.ptr, ra[0], dB.ptr, k, 0.0, dC.ptr, m, True)
            hashes.add(ctx.hash64(dC.ptr, m, n, m, True))
        lib.chase_hip_ctx_set_gemm_min_rounds(ctx.h, 0)
        print(f"op {op} {m} x {n} x {k}: {reps} launches, {len(hashes)} distinct hash(es)", flush=True)
        assert len(hashes) == 1
        for a in (dA, dB, dC):
            a.free()
    lib.chase_hip_ctx_set_phase(ctx.h, 0)
    for wl, reps in (("cfg2", 6), ("cfg3c", 3)):
        N, cplx, nev, nex = B.WORKLOADS[wl]
        dH = ctx.gen_clement(N, cplx, scale=B.MATRIX_SCALE / N, perturb=B.MATRIX_PERTURB, seed=42)
        res = []
        for rep in range(reps):
            s = Solver(ctx, None, nev, nex, h_on_device_ptr=dH.ptr, N=N, cplx=cplx)
            s.set(device_rng=1)
            st = s.solve()
            res.append((s.ritzv.copy(), s.resid().copy(), st["iterations"], st["filtered_vecs"], s.hash_V(nev) if hasattr(s, "hash_V") else s.V.copy().tobytes()))
            s.close()
        for r in res[1:]:
            assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and r[2:4] == res[0][2:4] and r[4] == res[0][4]
        print(f"{reps} {wl} solves bitwise identical:", res[0][2], res[0][3], float(res[0][1][:nev].max()), flush=True)
        dH.free()
