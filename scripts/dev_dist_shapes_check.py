"""development: the local GEMM shapes of the 4x2 / 2x2 / 2x1 grids at config 4 through the filter kernel (3M) against the
four-product kernel on the same operands (relative difference of the results)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check
with Context(0) as ctx:
    for (rows, cols) in ((16384, 32768), (32768, 32768), (32768, 65536)):        # H_loc of the 4x2, 2x2, 2x1 grids
        dA = ctx.empty((rows, cols), np.complex128)
        check(lib.chase_hip_fill_normal(ctx.h, 1, rows, cols, dA.ptr, rows, 0, 0, rows, 1), "fill")
        for op, m, k in (("N", rows, cols), ("C", cols, rows)):
            for n in (256, 133, 2560):
                dB = ctx.empty((k, n), np.complex128); dC = ctx.empty((m, n), np.complex128); dD = ctx.empty((m, n), np.complex128)
                check(lib.chase_hip_fill_normal(ctx.h, 1, k, n, dB.ptr, k, 0, 0, k, 2), "fill")
                check(lib.chase_hip_fill_normal(ctx.h, 1, m, n, dC.ptr, m, 0, 0, m, 3), "fill")
                check(lib.chase_hip_lacpy(ctx.h, 1, m, n, dC.ptr, m, dD.ptr, m), "lacpy")
                lib.chase_hip_ctx_set_phase(ctx.h, 1)
                lib.chase_hip_set_gemm3m(1)
                ctx.gemm(op, m, n, k, 0.5, dA.ptr, rows, dB.ptr, k, -0.25, dC.ptr, m, True)
                lib.chase_hip_set_gemm3m(0)
                ctx.gemm(op, m, n, k, 0.5, dA.ptr, rows, dB.ptr, k, -0.25, dD.ptr, m, True)
                lib.chase_hip_set_gemm3m(1)
                lib.chase_hip_ctx_set_phase(ctx.h, 0)
                C3 = dC.download(); C4 = dD.download()
                rel = np.abs(C3 - C4).max() / np.abs(C4).max()
                print(f"H_loc {rows}x{cols} op={op} m={m} k={k} n={n}: max|3M-4M|/max|C| = {rel:.2e}", flush=True)
                assert rel < 1e-12
                dB.free(); dC.free(); dD.free()
        dA.free()
print("ok")
