#!/usr/bin/env python3
"""Generates tests/golden/oracle_cfg3_fullsize_unperturbed_2x2.json: the CPU oracle (oracle/chase_oracle.py, numpy on the host
BLAS; the restatement of pChASECPU pinned in tests/test_oracle_pins.py) solving BASELINE configs[2]'s SHAPE AT FULL SIZE -
N = 32768 real symmetric, nev = 1024, nex = 256, tol 1e-10, deg 20, optimised degrees - in its pChASECPU form for the 2 x 2 BLOCK
grid (start vectors mt19937(1337 + grid row) per block of local rows, pchase_cpu.hpp:272-283; V2 refreshed by QR; Swap on both
blocks).  Matrix: the UNPERTURBED Clement-type matrix of the reference's solve tests (tests/chase_serial_solve.cpp:52-90)
scaled by 100 / N like bench.py's workloads - analytic spectrum {-100, -100 + 200/N, ...}; no perturbation, because drawing
5.4e8 normals from the oracle's mt19937 replay takes longer than the solve and the product's device generator draws its
perturbation from another stream anyway.  What the fixture is for: tests/test_gpu_fullsize.py compares the HIP grid Impl's
iteration and filtered-vector counts at FULL size with an independent implementation instead of with its own earlier runs.
Run time here: ~35 minutes on 8 cores, ~20 GB."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import chase_oracle as O  # noqa: E402

N, nev, nex = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32768, 1024, 256)
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "tests", "golden", "oracle_cfg3_fullsize_unperturbed_2x2.json")
# optional: another grid - number of grid rows and the block size of a block-cyclic row distribution (0 = block layout); only the
# grid ROWS matter to the oracle (they decide which rows a start-vector stream fills)
nprow = int(sys.argv[5]) if len(sys.argv) > 5 else 2
nb = int(sys.argv[6]) if len(sys.argv) > 6 else 0
cplx = len(sys.argv) > 7 and sys.argv[7] == "z"        # complex Hermitian twin (bench.py's cfg3c shape): 4 x the flops, 17 GB matrix
t0 = time.time()
H = O.clement(N, cplx, perturb=0)
H *= 100.0 / N
if nb == 0:
    assert N % nprow == 0
    rows = [np.arange(i * (N // nprow), (i + 1) * (N // nprow)) for i in range(nprow)]      # block layout: contiguous row blocks
else:
    g = np.arange(N)
    rows = [g[(g // nb) % nprow == i] for i in range(nprow)]                                   # block-cyclic (distMatrix.hpp:44-67)
k = O.OracleCPU(H, nev, nex, grid_rows=rows)
del H
st = O.solve(k)
lam = k.ritzv[:nev].copy()
exact = (100.0 / N) * (-N + 2.0 * np.arange(nev))
rec = {"what": "oracle (pChASECPU form, %d grid rows, %s) on the unperturbed Clement-type matrix x 100/N"
               % (nprow, "block layout" if nb == 0 else "block-cyclic nb = %d" % nb),
       "N": N, "nev": nev, "nex": nex, "complex": bool(cplx), "grid": "2x2" if (nprow, nb) == (2, 0) else "%dx*" % nprow,
       "layout": "block" if nb == 0 else "block-cyclic nb=%d" % nb, "grid_rows": nprow, "tol": k.config.tol, "deg": k.config.deg,
       "iterations": int(st["iterations"]), "filtered_vecs": int(st["filtered_vecs"]),
       "max_abs_dev_from_analytic": float(np.max(np.abs(np.sort(lam) - exact))),
       "max_resid": float(np.max(k.resid[:nev])), "lambda_first": lam[:4].tolist(), "lambda_last": lam[-2:].tolist(),
       "seconds": time.time() - t0}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
