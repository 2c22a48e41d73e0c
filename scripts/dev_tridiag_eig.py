"""development: heevd_gpu phases (CHASE_HIP_HEEVD_TIMING) with the host tridiagonal solver variants; residual / orthogonality"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CHASE_HIP_HEEVD_TIMING"] = "1"
from chase_amd.capi import Context, lib, check
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
kind = sys.argv[2] if len(sys.argv) > 2 else "random"
with Context(0) as ctx:
    if kind == "random":
        X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        A = np.asfortranarray(X + X.conj().T)
    else:                                   # prescribed spectrum with clusters: groups of 8 eigenvalues 1e-13 apart
        lam = np.repeat(np.arange(n // 8 + 1), 8)[:n] + 1e-13 * np.tile(np.arange(8), n // 8 + 1)[:n]
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
        A = np.asfortranarray((Q * lam[None, :]) @ Q.conj().T); A = np.asfortranarray((A + A.conj().T) / 2)
    for rep in range(3):
        dA = ctx.array(A); w = np.zeros(n)
        t = time.perf_counter()
        check(lib.chase_hip_heevd_gpu(ctx.h, 1, n, dA.ptr, n, w.ctypes.data), "heevd")
        dt = time.perf_counter() - t
        Z = dA.download(); dA.free()
    res = np.linalg.norm(A @ Z - Z * w[None, :]) / np.linalg.norm(A)
    G = Z.conj().T @ Z - np.eye(n)
    print(f"n={n} {kind} MRRR={os.environ.get('CHASE_HIP_TRIDIAG_MRRR')}: total {dt*1e3:.1f} ms resid {res:.1e} orth max {np.abs(G).max():.1e} fro/sqrt(n) {np.linalg.norm(G)/np.sqrt(n):.1e}", flush=True)
