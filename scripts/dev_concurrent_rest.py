"""experiment: the ragged rest of a width run CONCURRENTLY with the whole tiles (two contexts = two streams on one device)
instead of after them.  real N = 32768: n = 300 = 256 + 44, n = 428 = 384 + 44, n = 1200 = 1152 + 48"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check
N = 32768
cplx = len(sys.argv) > 1 and sys.argv[1] == "z"
if cplx: N = 16384
dt = np.complex128 if cplx else np.float64
BN = 64 if cplx else 128
with Context(0) as c1, Context(0) as c2:
    dA = c1.empty((N, N), dt)
    check(lib.chase_hip_fill_normal(c1.h, int(cplx), N, N, dA.ptr, N, 0, 0, N, 1), "fill")
    for n in ((300, 428, 1200) if not cplx else (133, 300, 620)):
        n1 = n - n % BN; rem = n - n1
        dB = c1.empty((N, n), dt); dC = c1.empty((N, n), dt)
        check(lib.chase_hip_fill_normal(c1.h, int(cplx), N, n, dB.ptr, N, 0, 0, N, 2), "fill")
        check(lib.chase_hip_fill_normal(c1.h, int(cplx), N, n, dC.ptr, N, 0, 0, N, 3), "fill")
        es = 16 if cplx else 8
        for c in (c1, c2): lib.chase_hip_ctx_set_phase(c.h, 1)
        def whole(ctx): ctx.gemm("N", N, n1, N, 0.5, dA.ptr, N, dB.ptr, N, 0.25, dC.ptr, N, cplx)
        def rest(ctx): ctx.gemm("N", N, rem, N, 0.5, dA.ptr, N, dB.ptr + n1 * N * es, N, 0.25, dC.ptr + n1 * N * es, N, cplx)
        def both_seq(): c1.gemm("N", N, n, N, 0.5, dA.ptr, N, dB.ptr, N, 0.25, dC.ptr, N, cplx)
        def timeit(fn, reps=5):
            fn(); c1.sync(); c2.sync()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            c1.sync(); c2.sync()
            return (time.perf_counter() - t) / reps * 1e3
        t_seq = timeit(both_seq)
        t_w = timeit(lambda: whole(c1)); t_r = timeit(lambda: rest(c1))
        res = {}
        for rounds in (0, 2, 4):
            lib.chase_hip_ctx_set_gemm_min_rounds(c1.h, rounds)
            res[rounds] = (timeit(lambda: (whole(c1), rest(c2))), timeit(lambda: (rest(c2), whole(c1))))
        lib.chase_hip_ctx_set_gemm_min_rounds(c1.h, 0)
        print(f"cplx={cplx} n={n} = {n1} + {rem}: one call {t_seq:.3f} ms; whole alone {t_w:.3f}, rest alone {t_r:.3f}; concurrent (whole first / rest first) "
              + "; ".join(f"K pieces x{r}: {a:.3f} / {b:.3f}" for r, (a, b) in res.items()), flush=True)
        dB.free(); dC.free()
