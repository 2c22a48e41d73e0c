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
pad = int(sys.argv[6]) if len(sys.argv) > 6 else 0          # extra rows in the leading dimensions (set-aliasing probe)
data = sys.argv[7] if len(sys.argv) > 7 else "normal"       # "clement": the bench's matrix (x 100/N, perturbation 1e-6) as A
M = int(os.environ.get("DEV_M", "0")) or N               # rows of A and C (op N only): one-round launches for L2 studies
with Context(0) as ctx:
    dt = np.complex128 if cplx else np.float64
    L = N + pad
    dA = ctx.empty((L, N), dt); dB = ctx.empty((L, n), dt); dC = ctx.empty((L, n), dt)
    if data == "clement":
        assert pad == 0
        check(lib.chase_hip_gen_clement(ctx.h, int(cplx), dA.ptr, N, N, N, N, N, 1, 0, 0, N, 1, 0, 0, 100.0 / N, 1e-6, 42), "gen")
    else:
        check(lib.chase_hip_fill_normal(ctx.h, int(cplx), L, N, dA.ptr, L, 0, 0, L, 1), "fill")
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), L, n, dB.ptr, L, 0, 0, L, 2), "fill")
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), L, n, dC.ptr, L, 0, 0, L, 3), "fill")
    lib.chase_hip_ctx_set_phase(ctx.h, 1)
    ctx.gemm(op, M, n, N, 0.5, dA.ptr, L, dB.ptr, L, 0.25, dC.ptr, L, cplx)
    ctx.timer_start()
    for _ in range(reps):
        ctx.gemm(op, M, n, N, 0.5, dA.ptr, L, dB.ptr, L, 0.25, dC.ptr, L, cplx)
    ms = ctx.timer_stop() / reps
    if os.environ.get("DEV_PER_LAUNCH"):            # sustained-load drift: time launches one by one
        per = []
        for _ in range(int(os.environ["DEV_PER_LAUNCH"])):
            ctx.timer_start()
            ctx.gemm(op, N, n, N, 0.5, dA.ptr, L, dB.ptr, L, 0.25, dC.ptr, L, cplx)
            per.append(ctx.timer_stop())
        F_ = 4 if cplx else 1
        tf = [2.0 * F_ * N * N * n / (t * 1e-3) / 1e12 for t in per]
        k = max(len(tf) // 6, 1)
        print("per-launch TFLOP/s, consecutive groups:", " ".join(f"{sum(tf[i:i+k])/len(tf[i:i+k]):.2f}" for i in range(0, len(tf), k)), flush=True)
    F = 4 if cplx else 1
    print(f"HEMM cplx={cplx} op={op} M={M} N={N} n={n} ld={L} tile_group={os.environ.get('CHASE_HIP_TILE_GROUP', 'default')} A={data}: {ms:.3f} ms {2.0*F*M*N*n/(ms*1e-3)/1e12:.2f} TFLOP/s", flush=True)
