"""development: rate of the local filter GEMM of the 4x2 grid at config 4 (H_loc 16384 x 32768) against the panel width of the
pipelined distributed HEMM (pchase_hip_impl.hpp: panel_)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check
rows, cols, nfull = 16384, 32768, 2560
with Context(0) as ctx:
    dA = ctx.empty((rows, cols), np.complex128)
    check(lib.chase_hip_fill_normal(ctx.h, 1, rows, cols, dA.ptr, rows, 0, 0, rows, 1), "fill")
    lib.chase_hip_ctx_set_phase(ctx.h, 1)
    lib.chase_hip_ctx_set_gemm_min_rounds(ctx.h, int(os.environ.get("DEV_MIN_ROUNDS", "0")))
    for op, m, k in (("N", rows, cols), ("C", cols, rows)):
        dB = ctx.empty((k, nfull), np.complex128); dC = ctx.empty((m, nfull), np.complex128)
        check(lib.chase_hip_fill_normal(ctx.h, 1, k, nfull, dB.ptr, k, 0, 0, k, 2), "fill")
        check(lib.chase_hip_fill_normal(ctx.h, 1, m, nfull, dC.ptr, m, 0, 0, m, 3), "fill")
        for w in (128, 256, 512, 1024, 2560):
            def sweep():
                for c in range(0, nfull, w):
                    ctx.gemm(op, m, min(w, nfull - c), k, 0.5, dA.ptr, rows, dB.ptr + 16 * c * k, k, -0.25, dC.ptr + 16 * c * m, m, True)
            sweep()
            ctx.timer_start()
            for _ in range(3): sweep()
            ms = ctx.timer_stop() / 3
            print(f"op={op} m={m} k={k}: 2560 columns in panels of {w:4d} (min_rounds {os.environ.get('DEV_MIN_ROUNDS', '0')}): {ms:7.2f} ms = {2.0*4*m*k*nfull/(ms*1e-3)/1e12:6.2f} TFLOP/s (model)", flush=True)
        dB.free(); dC.free()
    lib.chase_hip_ctx_set_phase(ctx.h, 0)
