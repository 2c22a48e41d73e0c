"""Single-kernel driver for profiling: filter-shaped HEMM at cfg2 size, random dense operands."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check
cplx = (sys.argv[1] == "z") if len(sys.argv) > 1 else True
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
n = int(sys.argv[3]) if len(sys.argv) > 3 else 640
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
op = sys.argv[5] if len(sys.argv) > 5 else "N"
with Context(0) as ctx:
    dt = np.complex128 if cplx else np.float64
    dA = ctx.empty((N, N), dt); dB = ctx.empty((N, n), dt); dC = ctx.empty((N, n), dt)
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), N, N, dA.ptr, N, 0, 0, N, 1), "fill")
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), N, n, dB.ptr, N, 0, 0, N, 2), "fill")
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), N, n, dC.ptr, N, 0, 0, N, 3), "fill")
    lib.chase_hip_ctx_set_phase(ctx.h, 1)
    ctx.gemm(op, N, n, N, 0.5, dA.ptr, N, dB.ptr, N, 0.25, dC.ptr, N, cplx)
    ctx.timer_start()
    for _ in range(reps):
        ctx.gemm(op, N, n, N, 0.5, dA.ptr, N, dB.ptr, N, 0.25, dC.ptr, N, cplx)
    ms = ctx.timer_stop() / reps
    F = 4 if cplx else 1
    print(f"HEMM cplx={cplx} op={op} N={N} n={n}: {ms:.3f} ms {2.0*F*N*N*n/(ms*1e-3)/1e12:.2f} TFLOP/s", flush=True)
