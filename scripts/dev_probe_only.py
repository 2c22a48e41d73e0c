"""bench.py's full-width probes without the solves (same Solver object, same matrix): python3 scripts/dev_probe_only.py [cfg4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from chase_amd.capi import Context, Solver
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N, cplx, nev, nex = B.WORKLOADS[wl]
ctx = Context(0)
dH = ctx.gen_clement(N, cplx, scale=B.MATRIX_SCALE / N, perturb=B.MATRIX_PERTURB, seed=42)
s = Solver(ctx, None, nev, nex, h_on_device_ptr=dH.ptr, N=N, cplx=cplx)
s.set(device_rng=1)
for three_m in (True, False, True):
    r = B.fullwidth_probe(s, ctx, N, cplx, nev + nex, three_m, reps=int(os.environ.get("REPS", "4")))
    print("3M" if three_m else "4M", f"{r['algorithmic']:.2f} model TFLOP/s, {r['achieved']:.2f} executed, {r['avg_launch_ms']:.1f} ms/launch", flush=True)
s.close(); dH.free(); ctx.close()
