import sys, time, os
import numpy as np
sys.path.insert(0, ".")
from chase_amd.capi import Context, lib, check
import ctypes as C
print("cpus", os.cpu_count(), "provider", lib.chase_hip_lapack_provider())
rng = np.random.default_rng(0)
with Context(0) as ctx:
    for n in (640, 1280, 2560):
        X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        A = np.asfortranarray(X + X.conj().T)
        for thr in (0, 8, 16, 32, 64, 128):
            if thr: lib.chase_hip_set_host_threads(thr)
            dA = ctx.array(A); w = np.zeros(n)
            t = time.perf_counter()
            check(lib.chase_hip_heevd(ctx.h, 1, n, dA.ptr, n, w.ctypes.data), "heevd")
            dt = time.perf_counter() - t
            dA.free()
            print(f"n={n} threads={thr or 'default'}: {dt*1e3:.1f} ms", flush=True)
            if n == 2560 and thr == 16: break
