"""HBM roofline of the streaming (O(N n)) kernels at BASELINE configs[3]'s shapes (N = 65536 complex, 2560 columns): each
kernel is timed with HIP events on the context stream (best of `reps` launches) and its ALGORITHMIC bytes (what the operation
must read + write once) are divided by that time.  Under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` the same
launches give the fabric bytes (scripts/r06_hbm_kernels.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md).
usage: python scripts/r06_hbm_kernels.py [N] [ncols] [reps] [out.json]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2560
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
out = sys.argv[4] if len(sys.argv) > 4 else None
PEAK, ACHIEVABLE = 8.0e12, 6.29e12
rows = []
lib.chase_hip_col_sumsq.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p]
lib.chase_hip_col_sumsq.restype = C.c_int


def timed(ctx, name, bytes_alg, fn, note=""):
    fn()
    ctx.sync()
    best = 1e30
    for _ in range(reps):
        ctx.timer_start()
        fn()
        best = min(best, ctx.timer_stop())
    tb = bytes_alg / (best * 1e-3) / 1e12
    rows.append({"kernel": name, "ms": best, "algorithmic_GB": bytes_alg / 1e9, "TBps": tb, "of_8.0": tb * 1e12 / PEAK,
                 "of_6.29": tb * 1e12 / ACHIEVABLE, "note": note})
    print(f"{name:34s} {best:9.3f} ms  {bytes_alg / 1e9:8.3f} GB  {tb:6.2f} TB/s  {tb * 1e12 / PEAK:5.2f} of 8.0  "
          f"{tb * 1e12 / ACHIEVABLE:5.2f} of 6.29  {note}", flush=True)


with Context(0) as ctx:
    Z = np.complex128
    E = 16
    X = ctx.empty((N, n), Z); Y = ctx.empty((N, n), Z); W = ctx.empty((N, n), Z)
    blk = N * n * E
    for i, a in enumerate((X, Y, W)):
        check(lib.chase_hip_fill_normal(ctx.h, 1, N, n, a.ptr, N, 0, 0, N, 11 + i), "fill")
    lam = np.linspace(-1.0, 1.0, n)
    res = np.zeros(n)
    dsc = ctx.empty((2 * n, 1), np.float64)
    dsc.upload(np.full((2 * n, 1), 0.5))
    gb = C.c_double()
    check(lib.chase_hip_hbm_copy_peak(ctx.h, 4 << 30, C.byref(gb)), "copy peak")
    rows.append({"kernel": "stream_copy probe (4 GiB, float4)", "TBps": gb.value / 1e3, "of_8.0": gb.value * 1e9 / PEAK,
                 "of_6.29": gb.value * 1e9 / ACHIEVABLE})
    print(f"stream copy probe: {gb.value / 1e3:.2f} TB/s", flush=True)
    dres = ctx.empty((n, 1), np.float64)
    timed(ctx, "resid_norms (W - lambda V)", 2 * blk,
          lambda: check(lib.chase_hip_resid_norms_dev(ctx.h, 1, N, n, W.ptr, N, X.ptr, N, lam.ctypes.data, dres.ptr, 0), "resid"),
          "Resd, every iteration")
    timed(ctx, "col_sumsq / nrm2", blk, lambda: check(lib.chase_hip_col_sumsq(ctx.h, 1, N, n, X.ptr, N, dres.ptr), "sumsq"))
    timed(ctx, "lacpy (copy2d)", 2 * blk, lambda: check(lib.chase_hip_lacpy(ctx.h, 1, N, n, X.ptr, N, Y.ptr, N), "lacpy"),
          "pass 1 of the reused first Chebyshev step; QR / RR copies")
    timed(ctx, "col_axpy (one complex scalar)", 3 * blk,
          lambda: check(lib.chase_hip_col_axpy(ctx.h, 1, N, n, dsc.ptr, 0, 0, 1.0, X.ptr, N, Y.ptr, N), "axpy"),
          "pass 2 of the reused first Chebyshev step")
    timed(ctx, "scale_rows", 2 * blk, lambda: check(lib.chase_hip_scale_rows(ctx.h, 1, N, n, Y.ptr, N, 0, 0.999), "scale"),
          "pass 3 of the reused first Chebyshev step")
    timed(ctx, "col_scal (device scalars)", 2 * blk,
          lambda: check(lib.chase_hip_col_scal(ctx.h, 1, N, n, dsc.ptr, 0, Y.ptr, N), "scal"))
    dd = ctx.empty((2 * n, 1), np.float64)
    timed(ctx, "col_dot", 2 * blk, lambda: check(lib.chase_hip_col_dot(ctx.h, 1, N, n, X.ptr, N, Y.ptr, N, dd.ptr), "dot"))
    timed(ctx, "col_dot, 4 columns (Lanczos)", 2 * N * 4 * E,
          lambda: check(lib.chase_hip_col_dot(ctx.h, 1, N, 4, X.ptr, N, Y.ptr, N, dd.ptr), "dot"),
          "latency-bound: 8 MB")
    timed(ctx, "col_axpy, 4 columns (Lanczos)", 3 * N * 4 * E,
          lambda: check(lib.chase_hip_col_axpy(ctx.h, 1, N, 4, dsc.ptr, 0, 1, -1.0, X.ptr, N, Y.ptr, N), "axpy"),
          "latency-bound: 12 MB")
    timed(ctx, "conj", 2 * blk, lambda: check(lib.chase_hip_conj(ctx.h, N, n, Y.ptr, N), "conj"))
    h = C.c_ulonglong()
    timed(ctx, "hash64 (+ 8-byte read-back)", blk, lambda: check(lib.chase_hip_hash64(ctx.h, 1, N, n, X.ptr, N, C.byref(h)), "hash"))
    timed(ctx, "fill_normal (Philox + Box-Muller)", blk,
          lambda: check(lib.chase_hip_fill_normal(ctx.h, 1, N, n, Y.ptr, N, 0, 0, N, 5), "fill"), "write-only, ALU-heavy")
    # packed triangle of the projected matrix
    A = ctx.empty((n, n), Z); P = ctx.empty((n * (n + 1) // 2, 1), Z)
    check(lib.chase_hip_fill_normal(ctx.h, 1, n, n, A.ptr, n, 0, 0, n, 21), "fill")
    tri = n * (n + 1) // 2 * E
    timed(ctx, "pack_upper (n = %d)" % n, 2 * tri, lambda: check(lib.chase_hip_pack_upper(ctx.h, 1, n, A.ptr, n, P.ptr), "pack"), "52 MB payload")
    timed(ctx, "unpack_upper + mirror", 2 * tri + tri, lambda: check(lib.chase_hip_unpack_upper(ctx.h, 1, n, P.ptr, A.ptr, n, 1), "unpack"))
    for a in (W, A, P):
        a.free()
    # the matrix generator at the size the bench uses it (68.7 GB written once)
    H = ctx.empty((N, N), Z)
    timed(ctx, "gen_clement (N x N, perturbed)", N * N * E,
          lambda: check(lib.chase_hip_gen_clement(ctx.h, 1, H.ptr, N, N, N, N, N, 1, 0, 0, N, 1, 0, 0, 100.0 / N, 1e-6, 42), "gen"),
          "write-only, Philox per element")
    timed(ctx, "shift_diag (N diagonal elements)", 2 * N * 8, lambda: check(lib.chase_hip_shift_diag(ctx.h, 1, N, H.ptr, N, 0.25), "shift"),
          "latency-bound: stride-N access")
if out:
    json.dump({"N": N, "ncols": n, "reps": reps, "rows": rows}, open(out, "w"), indent=1)
