"""development: the same pseudo-Hermitian operator sequence on the sequential Impl and on the grid Impl (1 x 1 grid)"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from chase_amd.capi import Context, PseudoSolver, lib, check
from chase_amd import dist as cd
from oracle import chase_oracle as O
from rank_threads import run_ranks
import conftest
H = conftest.read_ref_matrix("cdouble_random_BSE.bin", 200, 200, True)
N, nev, nex = 200, 20, 20
ne = nev + nex
rng = np.random.default_rng(3)
ritzVc = np.asfortranarray(rng.standard_normal((50, 8)) + 0j)
out = {}
def ops(s, getV, tag):
    s.set(numlanczos=10, lanczositer=50)
    s.Start(); s.initVecs(True); out[tag + "init"] = getV()
    s.QR(0, 1.0); out[tag + "qr"] = getV()
    r = s.Lanczos(50, 10); out[tag + "lan"] = getV(); out[tag + "ub"] = r[0]; out[tag + "theta"] = np.sort(np.asarray(r[1]).ravel())
    check(lib.chase_hip_op_lanczos_dos(s.h, 8, 50, ritzVc.ctypes.data), "dos"); out[tag + "dos"] = getV()
    for (a, b, g, off) in [(1e-3, 0.0, -0.2, 0), (2e-3, -0.3, -0.4, 0), (2e-3, -0.25, -0.4, 3), (2e-3, -0.25, -0.4, 3)]:
        s.HEMM_H2(ne, a, b, g, off)
    out[tag + "filt"] = getV()
    s.ApplyKconjugate(ne); out[tag + "kconj"] = getV()
    s.QR(0, 1e5); out[tag + "qr2"] = getV()
    s.RR(ne, 0); out[tag + "rr"] = getV(); out[tag + "ritz"] = s.ritzv.copy()
    out[tag + "resd"] = s.Resd(0)[:ne]
with Context(0) as ctx:
    s = PseudoSolver(ctx, H, nev, nex); ops(s, s.peek_v, "s_"); s.close()
def scen(ctx, grid, comm):
    dH = ctx.array(H)
    s = cd.DistPseudoSolver(ctx, grid, dH, N, nev, nex, True, 0, 0); ops(s, s.local_V, "g_"); s.close()
run_ranks(1, 1, scen)
for k in ("init", "qr", "lan", "dos", "filt", "kconj", "qr2", "rr"):
    a, b = out["s_" + k], out["g_" + k]
    sc = np.abs(a).max()
    d = np.minimum(np.abs(a - b).max(axis=0), np.abs(a + b).max(axis=0)) / sc
    print(f"{k:6s} max rel diff per column: {np.array2string(d, precision=0, max_line_width=400)}")
print("ub", out["s_ub"], out["g_ub"], "theta diff", np.abs(out["s_theta"] - out["g_theta"]).max())
print("ritz diff", np.abs(out["s_ritz"] - out["g_ritz"]).max(), "resd diff", np.abs(out["s_resd"] - out["g_resd"]).max())
