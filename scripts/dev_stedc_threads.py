"""development: host tridiagonal D&C time inside heevd_gpu vs provider thread count (n = 2560, 1280)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check
os.environ["CHASE_HIP_HEEVD_TIMING"] = "1"
rng = np.random.default_rng(0)
with Context(0) as ctx:
    for n in (1280, 2560):
        X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        A = np.asfortranarray(X + X.conj().T)
        for th in (8, 16, 32, 64, 128):
            lib.chase_hip_set_host_threads(th)
            for rep in range(2):
                dA = ctx.array(A); w = np.zeros(n)
                t = time.perf_counter()
                check(lib.chase_hip_heevd_gpu(ctx.h, 1, n, dA.ptr, n, w.ctypes.data), "heevd")
                dt = time.perf_counter() - t
                dA.free()
            print(f"n={n} threads={th}: total {dt*1e3:.1f} ms", flush=True)
