"""development: per-iteration trace of the cfg5 workload (1x1 grid) — locked counts, degrees, residual extremes."""
import os, sys, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
with socket.socket() as so:
    so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
import torch.distributed as dist
from chase_amd.capi import Context
from chase_amd import dist as cd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nev = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nex = int(sys.argv[3]) if len(sys.argv) > 3 else 25
off = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
dist.init_process_group("gloo")
ctx = Context(0)
grid = cd.Grid(ctx, 1, 1, 0, transport="rccl", pg=cd.make_process_groups(1, 1))
rl, cl = cd.Layout(N, 0, 1), cd.Layout(N, 0, 1)
dH = cd.gen_bse_local(ctx, N, True, rl, cl, 0, 0, dmin=1.0, dmax=11.0, offdiag=off)
s = cd.DistPseudoSolver(ctx, grid, dH, N, nev, nex, True)
s.set(device_rng=1, numlanczos=10, lanczositer=50)
st = s.solve(trace=True)
for l in s.trace():
    if l.startswith(("Lock", "filter", "QR", "bounds", "lanczos")):
        print(l)
print({k: st[k] for k in ("iterations", "locked", "filtered_vecs", "t_all", "t_filter", "lowerb", "upperb")})
r = s.resid()[:nev + nex]
print("resid sorted tail", np.sort(r)[-12:])
print("ritz head", s.ritzv[:6], "ritz around nev", s.ritzv[nev - 3:nev + 3])
s.close(); grid.close(); ctx.close(); dist.destroy_process_group()
