"""development: config-2 solve when H is handed over as a HOST buffer (the reference's constructor contract): the upload is
inside the solve (initVecs), so this is the PCIe-inclusive time quoted in DESIGN.md §6."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chase_amd.capi import Context, Solver
N, nev, nex = 16384, 512, 128
with Context(0) as ctx:
    dH = ctx.gen_clement(N, True, scale=100.0 / N, perturb=1e-6, seed=42)
    H = dH.download(); dH.free()
    s = Solver(ctx, H, nev, nex)
    s.set(device_rng=1)
    for rep in range(2):
        t = time.perf_counter(); st = s.solve(); dt = time.perf_counter() - t
        print(f"host-resident H, solve {rep}: {dt:.3f} s (t_init {st['t_init']:.3f} s incl. the {H.nbytes/1e9:.1f} GB upload), iterations {st['iterations']}", flush=True)
    s.close()
