"""development: the same operator sequence on the sequential Impl and on the grid Impl (1 x 1 grid), compared after every step - how the
LanczosDos / V2 difference of the two reference Impls was found (HISTORY.md section 5)"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from chase_amd.capi import Context, Solver, lib, check
from chase_amd import dist as cd
from oracle import chase_oracle as O
from rank_threads import run_ranks
N, nev, nex, cplx, deg = 301, 20, 10, False, 20
H = O.clement(N, cplx)
rng = np.random.default_rng(3)
ritzVc = np.asfortranarray(rng.standard_normal((24, 6)))
out = {}
def seqops(s, getV, tag):
    s.Start(); s.initVecs(True); s.QR(0, 1.0); s.Lanczos(24, 4)
    check(lib.chase_hip_op_lanczos_dos(s.h, 6, 24, ritzVc.ctypes.data), "dos"); out[tag + "dos"] = getV()
    c = 117.0
    s.Shift(-c)
    for (a, b) in [(0.01, 0.0), (0.02, -0.3), (0.02, -0.25), (0.015, -0.2), (0.016, -0.21), (0.017, -0.22)]:
        s.HEMM(30, a, b, 0)
    s.Shift(c, True); out[tag + "filt"] = getV()
    s.QR(0, 5.8e5); out[tag + "qr2"] = getV(); out[tag + "qrv"] = s.get("qr_variant")
    s.RR(30, 0); out[tag + "rr"] = getV(); out[tag + "ritz"] = s.ritzv.copy()
    out[tag + "resd"] = s.Resd(0)
    s.Swap(2, 7); s.Swap(7, 11); s.Swap(0, 29); s.Lock(4)
    s.Shift(-c)
    for (a, b) in [(0.01, 0.0), (0.02, -0.3), (0.02, -0.25), (0.02, -0.26)]:
        s.HEMM(26, a, b, 0)
    s.HEMM(20, 0.02, -0.25, 6); s.HEMM(20, 0.02, -0.25, 6)
    s.Shift(c, True); out[tag + "filt2"] = getV()
    s.QR(4, 3e10); out[tag + "qr3"] = getV(); out[tag + "qrv3"] = s.get("qr_variant")
    s.RR(26, 4); out[tag + "rr2"] = getV(); out[tag + "ritz2"] = s.ritzv.copy()
    out[tag + "resd2"] = s.Resd(4)
with Context(0) as ctx:
    s = Solver(ctx, H, nev, nex); s.set(deg=deg, device_rng=1)
    seqops(s, s.peek_v, "s_"); s.close()
def scen(ctx, grid, comm):
    dH = ctx.array(H)
    s = cd.DistSolver(ctx, grid, dH, N, nev, nex, cplx, 0, 0); s.set(deg=deg, device_rng=1)
    seqops(s, s.local_V, "g_"); s.close()
run_ranks(1, 1, scen)
for k in ("dos", "filt", "qr2", "rr", "filt2", "qr3", "rr2"):
    a, b = out["s_" + k], out["g_" + k]
    sc = np.abs(a).max()
    # eigenvector columns may differ by sign after RR
    d = np.minimum(np.abs(a - b).max(axis=0), np.abs(a + b).max(axis=0)) / sc
    print(f"{k:6s} max rel diff per column: {np.array2string(d, precision=1, max_line_width=300)}")
print("qr variants", out["s_qrv"], out["g_qrv"], out["s_qrv3"], out["g_qrv3"])
print("ritz diff", np.abs(out["s_ritz"] - out["g_ritz"]).max(), np.abs(out["s_ritz2"] - out["g_ritz2"]).max())
print("resd  seq", out["s_resd"][:8]); print("resd grid", out["g_resd"][:8])
print("resd2 seq", out["s_resd2"][:8]); print("resd2 grid", out["g_resd2"][:8])
