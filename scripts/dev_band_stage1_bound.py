#!/usr/bin/env python3
"""Round 5: settle the two-stage tridiagonalisation question with DATA.

Stage 1 of a two-stage Hermitian eigensolver reduces the dense n x n projected matrix to a band of width b by panel QR
factorisations (n/b - 1 panels of shrinking height m_j = n - (j + 1) b and width b) and two-sided compact-WY updates of the
trailing block.  This script times exactly those pieces at n = 2560, b = 64 with the kernels the library has - the panel
factorisation through chase_hip_houseqr (the Householder panel kernels of hhqr.hip) and the two-sided update as the three MFMA
GEMM shapes it consists of (W = A V, the small V^H W, the rank-2b update) - WITHOUT the glue that would make it a correct
reduction: a lower bound of what stage 1 would cost when built from the existing kernels.  Bar: <= 25 ms, else the idea is
dropped for good (the full heevd of n = 2560 takes 85 ms today, 63 of them the one-stage tridiagonalisation)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check  # noqa: E402


def main(n=2560, b=64, reps=3):
    ctx = Context(0)
    rng = np.random.default_rng(0)
    A = ctx.array(np.asfortranarray(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))))
    P = ctx.empty((n, b), np.complex128)
    V = ctx.empty((n, b), np.complex128)
    W = ctx.empty((n, b), np.complex128)
    S = ctx.empty((b, b), np.complex128)
    check(lib.chase_hip_fill_normal(ctx.h, 1, n, b, V.ptr, n, 0, 0, n, 3), "fill")
    heights = [n - (j + 1) * b for j in range(n // b - 1)]

    def panels():
        for m in heights:
            check(lib.chase_hip_lacpy(ctx.h, 1, m, b, A.ptr, n, P.ptr, n), "lacpy")
            check(lib.chase_hip_houseqr(ctx.h, 1, m, b, P.ptr, n), "houseqr")

    def updates():
        for m in heights:
            ctx.gemm("N", m, b, m, 1.0, A.ptr, n, V.ptr, n, 0.0, W.ptr, n, True)        # W = A V
            ctx.gemm("C", b, b, m, 1.0, V.ptr, n, W.ptr, n, 0.0, S.ptr, b, True)        # V^H W
            ctx.gemm("N", m, b, b, -0.5, V.ptr, n, S.ptr, b, 1.0, W.ptr, n, True)       # W -= 1/2 V (V^H W)
            # rank-2b update A -= W V^H + V W^H: two products with inner dimension b on the m x m trailing block ('C' on the
            # transposed operand is not in the ABI: the shapes and flops are those of A -= W * (b x m))
            ctx.gemm("N", m, m, b, -1.0, W.ptr, n, V.ptr, b, 1.0, A.ptr, n, True)
            ctx.gemm("N", m, m, b, -1.0, V.ptr, n, W.ptr, b, 1.0, A.ptr, n, True)

    out = {}
    for name, fn in (("panel factorisations (chase_hip_houseqr, %d panels of width %d)" % (len(heights), b), panels),
                     ("two-sided trailing updates (MFMA GEMM shapes)", updates)):
        fn()
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        out[name] = (time.perf_counter() - t) / reps * 1e3
    for k, v in out.items():
        print(f"{k:90s} {v:8.2f} ms")
    print(f"{'stage 1 lower bound from the existing kernels (n = %d, b = %d)' % (n, b):90s} {sum(out.values()):8.2f} ms   (bar: 25 ms)")
    ctx.close()


if __name__ == "__main__":
    main()
