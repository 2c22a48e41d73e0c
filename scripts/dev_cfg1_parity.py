import sys, numpy as np
sys.path.insert(0, '.')
from chase_amd.capi import Context, Solver
from oracle import chase_oracle as O
ctx = Context(0)
H = O.clement(4096, False, perturb=0)
s = Solver(ctx, H, 100, 40)
st = s.solve()
print("GPU mt19937 start:", st["iterations"], st["filtered_vecs"], float(np.max(s.resid()[:100])))
s.close()
