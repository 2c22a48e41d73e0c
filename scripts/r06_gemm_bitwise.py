"""The plane-fed 3M kernel (round 6) against round 2-5's loop (-DCHASE_M3_SPLANE=0 variant library): the same products must come out
BIT FOR BIT (the plane holds the same IEEE sums, accumulators are independent).  Run once per library, compare the printed hashes.
usage: [CHASE_HIP_LIB=...] python scripts/r06_gemm_bitwise.py"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check  # noqa: E402
shapes = [("N", 4096, 640, 4096), ("C", 4096, 640, 4096), ("N", 8192, 133, 8192), ("C", 8192, 133, 8192), ("N", 4096, 40, 4096),
          ("N", 1001, 100, 1001), ("C", 2560, 2560, 16384), ("N", 16384, 2560, 16384), ("N", 8193, 77, 8200), ("N", 128, 64, 8),
          ("N", 256, 64, 16), ("N", 256, 64, 24), ("N", 384, 200, 32)]
with Context(0) as ctx:
    for phase in (1, 2):
        lib.chase_hip_ctx_set_phase(ctx.h, phase)
        for (op, m, n, k) in shapes:
            ra = (m, k) if op == "N" else (k, m)
            dA = ctx.empty(ra, np.complex128); dB = ctx.empty((k, n), np.complex128); dC = ctx.empty((m, n), np.complex128)
            check(lib.chase_hip_fill_normal(ctx.h, 1, ra[0], ra[1], dA.ptr, ra[0], 0, 0, ra[0], 1), "fill")
            check(lib.chase_hip_fill_normal(ctx.h, 1, k, n, dB.ptr, k, 0, 0, k, 2), "fill")
            check(lib.chase_hip_fill_normal(ctx.h, 1, m, n, dC.ptr, m, 0, 0, m, 3), "fill")
            ctx.gemm(op, m, n, k, 0.5 - 0.25j, dA.ptr, ra[0], dB.ptr, k, 0.25 + 0.5j, dC.ptr, m, True)
            h = ctx.hash64(dC.ptr, m, n, m, True)
            err = ""
            if m * n * k <= 4096 * 640 * 4096:
                A = dA.download(); B = dB.download()
                C0 = np.empty((m, n), np.complex128, order="F")
                g = ctx.empty((m, n), np.complex128)
                check(lib.chase_hip_fill_normal(ctx.h, 1, m, n, g.ptr, m, 0, 0, m, 3), "fill")
                C0 = g.download(); g.free()
                ref = (0.5 - 0.25j) * ((A if op == "N" else A.conj().T) @ B) + (0.25 + 0.5j) * C0
                bound = (np.abs(A if op == "N" else A.T) @ np.abs(B)).max()
                err = " max_err/(sum|a||b|) = %.2e" % (np.abs(dC.download() - ref).max() / bound)
            print(f"phase {phase} op {op} m {m} n {n} k {k}: hash {h:016x}{err}", flush=True)
            for a in (dA, dB, dC):
                a.free()
