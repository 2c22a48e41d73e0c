"""Development probe (runs on the GPU box): GEMM correctness vs numpy on odd shapes + first throughput numbers."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from chase_amd.capi import Context

rng = np.random.default_rng(0)
out = {}
with Context(0) as ctx:
    print(ctx.info(), flush=True)
    out["mfma_f64_peak_tflops"] = ctx.mfma_f64_peak()
    out["hbm_copy_gbps"] = ctx.hbm_copy_peak(1 << 30)
    print(out, flush=True)

    def rnd(shape, cplx):
        a = rng.standard_normal(shape)
        if cplx:
            a = a + 1j * rng.standard_normal(shape)
        return np.asfortranarray(a)

    worst = 0.0
    for cplx in (False, True):
        for op in ("N", "C"):
            for (m, n, k) in [(128, 128, 64), (256, 128, 128), (130, 67, 45), (1, 1, 1), (17, 300, 1000),
                              (300, 17, 33), (512, 192, 777), (64, 64, 4096), (200, 140, 5000)]:
                A = rnd((m, k) if op == "N" else (k, m), cplx)
                B = rnd((k, n), cplx)
                Cm = rnd((m, n), cplx)
                alpha = (0.7 - 0.3j) if cplx else 0.7
                beta = (-0.4 + 0.2j) if cplx else -0.4
                for bt in (0.0, beta):
                    dA, dB, dC = ctx.array(A), ctx.array(B), ctx.array(Cm)
                    ctx.gemm(op, m, n, k, alpha, dA.ptr, dA.ld, dB.ptr, dB.ld, bt, dC.ptr, dC.ld, cplx)
                    got = dC.download()
                    opA = A if op == "N" else A.conj().T
                    ref = alpha * (opA @ B) + bt * Cm
                    scale = np.abs(opA) @ np.abs(B) * abs(alpha) + abs(bt) * np.abs(Cm) + 1e-300
                    err = np.max(np.abs(got - ref) / scale)
                    worst = max(worst, err)
                    flag = "" if err < 1e-14 else "  <-- BAD"
                    print(f"cplx={cplx} op={op} m={m} n={n} k={k} beta={bt}: rel err {err:.2e}{flag}", flush=True)
                    for d in (dA, dB, dC):
                        d.free()
    out["worst_rel_err"] = worst

    # throughput: filter-shaped products
    for (cplx, N, n) in [(False, 8192, 1280), (True, 8192, 640), (True, 16384, 640), (False, 16384, 1280),
                         (True, 16384, 256), (True, 16384, 64)]:
        dt = np.complex128 if cplx else np.float64
        dA = ctx.empty((N, N), dt); dB = ctx.empty((N, n), dt); dC = ctx.empty((N, n), dt)
        # fill with random data on host in slabs (random data, not zeros: DVFS)
        blk = rnd((N, 256), cplx)
        for j in range(0, N, 256):
            import ctypes
            from chase_amd.capi import lib, check
            check(lib.chase_hip_memcpy_h2d(ctx.h, dA.offset(j), blk.ctypes.data, blk.nbytes), "h2d")
        dB.upload(rnd((N, n), cplx)); dC.upload(rnd((N, n), cplx))
        for op in ("N", "C"):
            for rep in range(2):
                ctx.timer_start()
                iters = 3
                for _ in range(iters):
                    ctx.gemm(op, N, n, N, 0.5, dA.ptr, N, dB.ptr, N, 0.25, dC.ptr, N, cplx)
                ms = ctx.timer_stop() / iters
            F = 4 if cplx else 1
            tf = 2.0 * F * N * N * n / (ms * 1e-3) / 1e12
            print(f"HEMM cplx={cplx} op={op} N={N} n={n}: {ms:.3f} ms  {tf:.2f} TFLOP/s", flush=True)
            out[f"hemm_{'z' if cplx else 'd'}_{op}_N{N}_n{n}_tflops"] = tf
        for d in (dA, dB, dC):
            d.free()
    # Gram-shaped (split-K)
    for (cplx, N, n) in [(True, 16384, 640), (False, 32768, 1280)]:
        dt = np.complex128 if cplx else np.float64
        dV = ctx.array(rnd((N, n), cplx)); dG = ctx.empty((n, n), dt)
        for rep in range(2):
            ctx.timer_start()
            ctx.gemm("C", n, n, N, 1.0, dV.ptr, N, dV.ptr, N, 0.0, dG.ptr, n, cplx)
            ms = ctx.timer_stop()
        F = 4 if cplx else 1
        print(f"GRAM cplx={cplx} N={N} n={n}: {ms:.3f} ms {2.0*F*N*n*n/(ms*1e-3)/1e12:.2f} TFLOP/s", flush=True)
        dV.free(); dG.free()
import os
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/dev_gemm_check.json", "w"), indent=1)
print(json.dumps(out))
