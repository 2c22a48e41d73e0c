import sys, time, os
import numpy as np
sys.path.insert(0, ".")
from chase_amd.capi import Context, lib, check
import scipy.linalg as sla
rng = np.random.default_rng(0)
with Context(0) as ctx:
    for cplx in (True,):
        for n in (640, 640, 1280, 2560):
            if n == 2560 and not cplx: continue
            X = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
            A = np.asfortranarray(X + X.conj().T)
            for which in ("gpu", "host"):
                dA = ctx.array(A); w = np.zeros(n)
                fn = lib.chase_hip_heevd_gpu if which == "gpu" else lib.chase_hip_heevd
                if which == "host": os.environ["CHASE_HIP_HEEVD_GPU_MIN"] = "0"
                t = time.perf_counter()
                rc = fn(ctx.h, int(cplx), n, dA.ptr, n, w.ctypes.data)
                dt = time.perf_counter() - t
                assert rc == 0, lib.chase_hip_last_error()
                Z = dA.download(); dA.free()
                if n <= 1280:
                    wref = sla.eigvalsh(A)
                    err = np.abs(w - wref).max() / np.abs(wref).max()
                else:
                    err = -1
                res = np.linalg.norm(A @ Z - Z * w[None, :]) / np.linalg.norm(A)
                orth = np.linalg.norm(Z.conj().T @ Z - np.eye(n)) / np.sqrt(n)
                print(f"cplx={cplx} n={n} {which}: {dt*1e3:.1f} ms  eig err {err:.1e} resid {res:.1e} orth {orth:.1e}", flush=True)
