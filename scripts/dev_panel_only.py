"""Single-shape driver for profiling the panel products of the pipelined distributed HEMM: C (m x w) = op(A) B for a LOCAL
block of H (e.g. the 4x2 grid at config 4: H_loc 16384 x 32768; op C: m = 32768, k = 16384; op N: m = 16384, k = 32768),
filter-phase kernel symbol, optional K-piece granularity of shared-chip launches.
usage: dev_panel_only.py <d|z> <op N|C> <rows of H_loc> <cols of H_loc> <panel width> <min_rounds> [reps]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check
cplx = sys.argv[1] == "z"
op, rows, cols, w, rounds = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
m, k = (rows, cols) if op == "N" else (cols, rows)
with Context(0) as ctx:
    dt = np.complex128 if cplx else np.float64
    dA = ctx.empty((rows, cols), dt); dB = ctx.empty((k, w), dt); dC = ctx.empty((m, w), dt)
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), rows, cols, dA.ptr, rows, 0, 0, rows, 1), "fill")
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), k, w, dB.ptr, k, 0, 0, k, 2), "fill")
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), m, w, dC.ptr, m, 0, 0, m, 3), "fill")
    lib.chase_hip_ctx_set_phase(ctx.h, 1)
    lib.chase_hip_ctx_set_gemm_min_rounds(ctx.h, rounds)
    ctx.gemm(op, m, w, k, 0.5, dA.ptr, rows, dB.ptr, k, 0.25, dC.ptr, m, cplx)
    ctx.timer_start()
    for _ in range(reps):
        ctx.gemm(op, m, w, k, 0.5, dA.ptr, rows, dB.ptr, k, 0.25, dC.ptr, m, cplx)
    ms = ctx.timer_stop() / reps
    F = 4 if cplx else 1
    alg = (rows * cols + (m + k) * w * 2) * (16 if cplx else 8)
    print(f"panel cplx={cplx} op={op} H_loc={rows}x{cols} w={w} min_rounds={rounds}: {ms:.3f} ms "
          f"{2.0*F*m*k*w/(ms*1e-3)/1e12:.2f} TFLOP/s (model), algorithmic bytes {alg/1e9:.3f} GB", flush=True)
