import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from chase_amd.capi import Context, Solver, lib, check
from oracle import chase_oracle as O
with Context(0) as ctx:
    # generator sanity: Hermitian, matches the numpy Clement when unperturbed
    for cplx in (False, True):
        d = ctx.gen_clement(500, cplx, 1.0, 0.0).download()
        print("unperturbed equal:", np.array_equal(d, O.clement(500, cplx, perturb=0)))
        d = ctx.gen_clement(500, cplx, 0.2, 1e-6).download()
        print("hermitian:", np.array_equal(d, d.conj().T), "pert std", (d - 0.2*O.clement(500, cplx, perturb=0))[2:,2:][np.triu_indices(498,1)].std()/0.2)
        X = ctx.empty((1000, 64), np.complex128 if cplx else np.float64)
        check(lib.chase_hip_fill_normal(ctx.h, int(cplx), 1000, 64, X.ptr, 1000, 0, 0, 1000, 1337), "fill")
        x = X.download(); print("normal mean/std", x.mean(), x.real.std(), (x.imag.std() if cplx else 0))
    N, nev, nex, cplx = 16384, 512, 128, True
    dH = ctx.gen_clement(N, cplx, 100.0/N, 1e-6)
    s = Solver(ctx, None, nev, nex, h_on_device_ptr=dH.ptr, N=N, cplx=True); s.set(device_rng=1)
    for rep in range(2):
        st = s.solve(trace=True); print(st)
    print([l for l in s.trace() if l.startswith(("Lock","filter"))])
