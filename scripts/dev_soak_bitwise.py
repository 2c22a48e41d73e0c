import sys, numpy as np
sys.path.insert(0, ".")
import bench as B
from chase_amd.capi import Context, Solver
N, cplx, nev, nex = B.WORKLOADS["cfg2"]
ctx = Context(0)
dH = ctx.gen_clement(N, cplx, scale=B.MATRIX_SCALE / N, perturb=B.MATRIX_PERTURB, seed=42)
res = []
for rep in range(4):
    s = Solver(ctx, None, nev, nex, h_on_device_ptr=dH.ptr, N=N, cplx=cplx)
    s.set(device_rng=1)
    st = s.solve()
    res.append((s.ritzv.copy(), s.resid().copy(), st["iterations"], st["filtered_vecs"], s.V.copy()))
    s.close()
for r in res[1:]:
    assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and r[2:4] == res[0][2:4] and np.array_equal(r[4], res[0][4])
print("4 config-2 solves bitwise identical:", res[0][2], res[0][3], float(res[0][1][:nev].max()))
