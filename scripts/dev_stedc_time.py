"""timing of chase_hip_stedc (device divide & conquer) against the host dstedc on the tridiagonal matrices of a random and of a
Clement-type projected problem; CHASE_HIP_HEEVD_TIMING=1 prints the stages of chase_hip_heevd_gpu"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check
import scipy.linalg as sla
rng = np.random.default_rng(0)
with Context(0) as ctx:
    lib.chase_hip_host_lapack_warmup()
    for n in (640, 1280, 2560):
        for name, (d, e) in {"random": (rng.standard_normal(n), rng.standard_normal(n - 1)),
                             "toeplitz": (2 * np.ones(n), np.ones(n - 1))}.items():
            w = np.zeros(n); dZ = ctx.empty((n, n), np.float64)
            for rep in range(3):
                t = time.perf_counter()
                check(lib.chase_hip_stedc(ctx.h, n, d.ctypes.data, e.ctypes.data, w.ctypes.data, dZ.ptr, n), "stedc")
                dt = time.perf_counter() - t
            Z = dZ.download(); dZ.free()
            T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
            res = np.abs(T @ Z - Z * w[None, :]).max() / (n * 1.1e-16 * max(np.abs(d).max(), np.abs(e).max()))
            orth = np.abs(Z.T @ Z - np.eye(n)).max() / (n * 1.1e-16)
            print(f"stedc n={n} {name}: device D&C {dt*1e3:.2f} ms (resid {res:.2f} n eps, orth {orth:.2f} n eps)", flush=True)
    # whole heevd
    for n in (1280, 2560):
        X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        A = np.asfortranarray(X + X.conj().T)
        for rep in range(4):
            os.environ["CHASE_HIP_STEDC_GPU_MIN"] = "512"
            dA = ctx.array(A); w = np.zeros(n)
            t = time.perf_counter()
            check(lib.chase_hip_heevd_gpu(ctx.h, 1, n, dA.ptr, n, w.ctypes.data), "heevd_gpu")
            dt = time.perf_counter() - t
            Z = dA.download(); dA.free()
        res = np.linalg.norm(A @ Z - Z * w[None, :]) / np.linalg.norm(A)
        orth = np.linalg.norm(Z.conj().T @ Z - np.eye(n)) / np.sqrt(n)
        print(f"heevd_gpu complex n={n}: {dt*1e3:.1f} ms resid {res:.1e} orth {orth:.1e}", flush=True)
