#!/usr/bin/env python3
"""Generates tests/golden/oracle_cfg5_fullsize_synthetic_bse_4x2.json: the CPU oracle (oracle/chase_oracle.py OraclePseudoCPU in its
pChASECPU form for 4 grid rows, block layout; solve_pseudo = algorithm/algorithm.inc:1834-2220) on BASELINE configs[4]'s SHAPE AT
FULL SIZE - N = 32768 complex pseudo-Hermitian (Bethe-Salpeter structure), nev = 256, nex = 64, numLanczos 10, lanczosIter 50
like the reference's BSE tests / examples/5_bse_benchmark - with the reference's start vectors (mt19937(1337 + grid row) per block of
local rows, global lower half damped).  Matrix: oracle.synthetic_bse_block (diagonal like bench.py's device generator, off-diagonal
entries from a hash of the index pair, so that the ranks of the GPU test can build the SAME matrix shard by shard).
What the fixture is for: tests/test_gpu_fullsize.py compares the HIP grid Impl's full-size pseudo-Hermitian solve with an independent
implementation.  Run time here: ~2 hours on 8 cores, ~40 GB."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import chase_oracle as O  # noqa: E402

N, nev, nex = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32768, 256, 64)
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "tests", "golden", "oracle_cfg5_fullsize_synthetic_bse_4x2.json")
nprow = int(sys.argv[5]) if len(sys.argv) > 5 else 4
BSE = {"dmin": 1.0, "dmax": 11.0, "offdiag": 1e-3}
t0 = time.time()
H = np.empty((N, N), dtype=np.complex128, order="F")
cols = np.arange(N)
for r0 in range(0, N, 1024):
    rows = np.arange(r0, min(N, r0 + 1024))
    H[rows, :] = O.synthetic_bse_block(rows, cols, N, **BSE)
assert N % nprow == 0
grid_rows = [np.arange(i * (N // nprow), (i + 1) * (N // nprow)) for i in range(nprow)]
k = O.OraclePseudoCPU(H, nev, nex, grid_rows=grid_rows)
del H
k.config.num_lanczos, k.config.lanczos_iter = 10, 50
st = O.solve_pseudo(k)
lam = k.ritzv[:nev].copy()
rec = {"what": "oracle (pChASECPU pseudo-Hermitian form, %d grid rows, block layout) on oracle.synthetic_bse_block" % nprow,
       "N": N, "nev": nev, "nex": nex, "grid_rows": nprow, "layout": "block", "bse": BSE, "num_lanczos": 10, "lanczos_iter": 50,
       "iterations": int(st["iterations"]), "filtered_vecs": int(st["filtered_vecs"]),
       "max_resid": float(np.max(k.resid[:nev])), "lambda_first": lam[:6].tolist(), "lambda_last": lam[-2:].tolist(),
       "lambda_sum": float(np.sum(lam)), "seconds": time.time() - t0}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
