#!/usr/bin/env python3
"""Generates tests/golden/oracle_cfg2_fullsize.json: the CPU oracle (oracle/chase_oracle.py, numpy on the host BLAS; the
restatement of ChASECPU pinned in tests/test_oracle_pins.py) solving BASELINE configs[1] AT FULL SIZE - N = 16384 complex
Hermitian, nev = 512, nex = 128, tol 1e-10, deg 20, optimised degrees - in its ChASECPU (single-process) form: one column-major
fill of the start block from mt19937(1337) (Impl/chase_cpu/chase_cpu.hpp:296-309).  Two matrices:
  * "unperturbed": the Clement-type matrix of the reference's solve tests (tests/chase_serial_solve.cpp:52-90) scaled by
    100 / N like bench.py's workloads - analytic spectrum {-100, -100 + 200/N, ...};
  * "perturbed": the same plus bench.py's dense Hermitian 1e-6 N(0,1) perturbation, drawn by the oracle's mt19937 replay of
    the reference tests' generator (O.clement(N, True)) - the matrix tests/test_gpu_solve.py hands the single-GPU Impl.
What the fixture is for: tests/test_gpu_fullsize.py compares the single-GPU HIP Impl's iteration and filtered-vector counts at
FULL size with an independent implementation (the assertions of tests/chase_serial_solve.cpp:36-140 plus the counts), and
tests/test_gpu_bench.py reads the bench workload's expected counts from here instead of from a literal.
Run time here: ~15-25 minutes per matrix on 8 cores, ~10 GB."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import chase_oracle as O  # noqa: E402

N, nev, nex = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16384, 512, 128)
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "tests", "golden", "oracle_cfg2_fullsize.json")
which = sys.argv[5].split(",") if len(sys.argv) > 5 else ["unperturbed"]
rec = json.load(open(out)) if os.path.exists(out) else {}
rec.update(N=N, nev=nev, nex=nex, complex=True, form="ChASECPU (one process, start block mt19937(1337) column-major)")
for name in which:
    t0 = time.time()
    H = O.clement(N, True, perturb=0) if name == "unperturbed" else O.clement(N, True)
    H *= 100.0 / N
    k = O.OracleCPU(H, nev, nex)
    del H
    st = O.solve(k)
    lam = k.ritzv[:nev].copy()
    exact = (100.0 / N) * (-N + 2.0 * np.arange(nev))
    rec[name] = {"what": "oracle (ChASECPU form) on the %s Clement-type matrix x 100/N" % name,
                 "tol": k.config.tol, "deg": k.config.deg,
                 "iterations": int(st["iterations"]), "filtered_vecs": int(st["filtered_vecs"]),
                 "max_abs_dev_from_analytic": float(np.max(np.abs(np.sort(lam) - exact))),
                 "max_resid": float(np.max(k.resid[:nev])), "lambda_first": lam[:4].tolist(), "lambda_last": lam[-2:].tolist(),
                 "lambda_sum": float(np.sum(lam)), "seconds": time.time() - t0}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec[name]), flush=True)
