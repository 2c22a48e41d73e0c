"""development: accuracy of the 3M filter kernel vs the 4M kernel against a long-double reference (HISTORY.md §3.1c)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, lib

def ld_matmul(A, B):
    Ar, Ai = A.real.astype(np.longdouble), A.imag.astype(np.longdouble)
    Br, Bi = B.real.astype(np.longdouble), B.imag.astype(np.longdouble)
    return (Ar @ Br - Ai @ Bi), (Ar @ Bi + Ai @ Br)

ctx = Context(0)
rng = np.random.default_rng(3)
m, k, n = 256, 4096, 64
for label, imag_scale in (("generic complex", 1.0), ("nearly real (imag 1e-8)", 1e-8)):
    A = rng.standard_normal((m, k)) + 1j * imag_scale * rng.standard_normal((m, k))
    B = rng.standard_normal((k, n)) + 1j * imag_scale * rng.standard_normal((k, n))
    Rr, Ri = ld_matmul(A, B)
    out = {}
    for phase, name in ((0, "4M"), (1, "3M")):
        lib.chase_hip_ctx_set_phase(ctx.h, phase)
        dA, dB, dC = ctx.array(np.asfortranarray(A)), ctx.array(np.asfortranarray(B)), ctx.array(np.zeros((m, n), dtype=complex, order="F"))
        ctx.gemm("N", m, n, k, 1.0, dA.ptr, m, dB.ptr, k, 0.0, dC.ptr, m, True)
        C = dC.download()
        er = np.abs(C.real.astype(np.longdouble) - Rr); ei = np.abs(C.imag.astype(np.longdouble) - Ri)
        nrm = float(np.sqrt((Rr ** 2 + Ri ** 2).sum()))
        out[name] = (float(np.sqrt((er ** 2 + ei ** 2).sum())) / nrm, float(er.max() / np.abs(Rr).max()), float(ei.max() / np.abs(Ri).max()))
    lib.chase_hip_ctx_set_phase(ctx.h, 0)
    for name, (fro, rre, rim) in out.items():
        print(f"{label:26s} {name}: ||err||_F/||C||_F = {fro:.2e}   max|err_re|/max|re| = {rre:.2e}   max|err_im|/max|im| = {rim:.2e}")
ctx.close()
