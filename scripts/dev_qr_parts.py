"""Times the parts of one CholQR pass at a bench size: Gram (herk), potrf, trsm.  usage: dev_qr_parts.py [N n cplx]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib, check
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2560
cplx = (sys.argv[3] == "z") if len(sys.argv) > 3 else True
dt = np.complex128 if cplx else np.float64
with Context(0) as ctx:
    V = ctx.empty((N, n), dt); A = ctx.empty((n, n), dt); A0 = ctx.empty((n, n), dt)
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), N, n, V.ptr, N, 0, 0, N, 5), "fill")
    def timed(fn, reps=3):
        fn(); ctx.sync(); ctx.timer_start()
        for _ in range(reps): fn()
        return ctx.timer_stop() / reps
    t_herk = timed(lambda: check(lib.chase_hip_herk(ctx.h, int(cplx), n, N, V.ptr, N, A0.ptr, n), "herk"))
    def potrf():
        check(lib.chase_hip_lacpy(ctx.h, int(cplx), n, n, A0.ptr, n, A.ptr, n), "lacpy")
        assert lib.chase_hip_potrf_upper(ctx.h, int(cplx), n, A.ptr, n) == 0
    t_potrf = timed(potrf)
    t_copy = timed(lambda: check(lib.chase_hip_lacpy(ctx.h, int(cplx), n, n, A0.ptr, n, A.ptr, n), "lacpy"))
    potrf()
    t_trsm = timed(lambda: check(lib.chase_hip_trsm_right_upper(ctx.h, int(cplx), N, n, A.ptr, n, V.ptr, N), "trsm"), reps=2)
    F = 4 if cplx else 1
    print(f"N={N} n={n} cplx={cplx}: herk {t_herk:.1f} ms ({0.55*2*F*N*n*n/t_herk/1e9:.1f} TF/s on the trapezoid), potrf {t_potrf - t_copy:.1f} ms, "
          f"trsm {t_trsm:.1f} ms ({F*N*n*n/t_trsm/1e9:.1f} TF/s on the triangle)")
