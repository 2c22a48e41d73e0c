"""development: does the GPU tridiagonalisation slow down after a long MFMA-heavy phase (clock / power state)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check
rng = np.random.default_rng(0)
os.environ["CHASE_HIP_HEEVD_TIMING"] = "1"
with Context(0) as ctx:
    n = 640
    X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A = np.asfortranarray(X + X.conj().T)
    def heevd(tag):
        dA = ctx.array(A); w = np.zeros(n)
        t = time.perf_counter()
        check(lib.chase_hip_heevd_gpu(ctx.h, 1, n, dA.ptr, n, w.ctypes.data), "heevd")
        print(tag, f"{(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
        dA.free()
    heevd("cold"); heevd("warm")
    N, nc = 16384, 640
    dH = ctx.empty((N, N), np.complex128); dB = ctx.empty((N, nc), np.complex128); dC = ctx.empty((N, nc), np.complex128)
    check(lib.chase_hip_fill_normal(ctx.h, 1, N, N, dH.ptr, N, 0, 0, N, 1), "fill")
    check(lib.chase_hip_fill_normal(ctx.h, 1, N, nc, dB.ptr, N, 0, 0, N, 2), "fill")
    heevd("after alloc of 4.6 GB")
    lib.chase_hip_ctx_set_phase(ctx.h, 1)
    for _ in range(30):
        ctx.gemm("N", N, nc, N, 0.5, dH.ptr, N, dB.ptr, N, 0.0, dC.ptr, N, True)
    lib.chase_hip_ctx_set_phase(ctx.h, 0)
    ctx.sync()
    heevd("right after 30 HEMMs"); heevd("again")
    time.sleep(1.0)
    heevd("after 1 s idle")
