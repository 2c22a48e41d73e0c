"""SURVEY.md §8 row A0 / (b) pinned to the REAL reference driver.

tests/golden/driver_trace_*.txt are runs of the reference's own chase::Solve (algorithm/algorithm.inc:1376-1788, compiled
from /root/reference by tests/golden/make_driver_traces.sh) on a naive CPU kernel deriving from the reference's
chase::ChaseBase<double>: iteration count, filtered-vector count, eigenpairs and EVERY virtual call with its scalar
arguments.  Checked here, without a GPU:
  * the product's C++ driver (chase_amd/host/algorithm.hpp) on the bit-identical kernel issues exactly that call sequence;
  * the Python oracle (own numerics: LAPACK instead of Jacobi / Gram-Schmidt) issues the same driver-level calls;
  * in the build container: the committed files still equal what the reference driver produces, and the four Impl classes
    instantiate on the reference's own chase::ChaseBase<T> (all 36 virtuals overridden: not abstract).
The GPU twin (whole trace of the HIP solver against the same files) is tests/test_gpu_solve.py."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import golden_traces as G
from oracle import chase_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path_factory.mktemp("hd") / "host_driver_harness"
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", str(exe), os.path.join(ROOT, "tests", "host_driver_harness.cpp")],
                   check=True, cwd=ROOT)
    return str(exe)


@pytest.mark.parametrize("name", ["clement256", "clement256_fix", "clement512", "clement1001"])
def test_own_driver_issues_the_reference_drivers_calls(harness, name):
    N, nev, nex, deg, opt, perturb = G.CASES[name]
    want = G.load(name)
    out = subprocess.run([harness, str(N), str(nev), str(nex), str(deg), str(opt), repr(perturb)], check=True,
                         capture_output=True, text=True, timeout=900).stdout.splitlines()
    got = G.parse_run(out)
    assert got["iterations"] == want["iterations"] == got["stats_iterations"]
    assert got["filtered_vecs"] == want["filtered_vecs"] == got["stats_filtered_vecs"]
    # same kernel arithmetic on both sides: every call, every scalar argument (Shift, Swap, HEMM alpha/beta, QR cond ...)
    G.assert_same_calls(got["calls"], want["calls"], 1e-12, "own C++ driver")
    assert np.max(np.abs(np.array(got["lam"]) - np.array(want["lam"]))) < 1e-12 * N
    # the driver-side trace the C ABI exposes (chase_hip_solver_trace) is the CORE subset of the kernel-side call list
    G.assert_same_calls([t for t in got["trace"] if t.split()[0] not in ("bounds", "filter")], G.core(want["calls"]), 1e-5,
                        "driver-side trace")


@pytest.mark.parametrize("name", ["clement256", "clement256_fix", "clement512", "clement1001", "clement1200"])
def test_oracle_issues_the_reference_drivers_calls(name):
    N, nev, nex, deg, opt, perturb = G.CASES[name]
    want = G.load(name)
    k = O.OracleCPU(O.clement(N, False, perturb=perturb), nev, nex)
    k.config.deg, k.config.opt = deg, bool(opt)
    tr = []
    so = O.solve(k, tr)
    assert so["iterations"] == want["iterations"]
    assert so["filtered_vecs"] == want["filtered_vecs"]
    G.assert_same_calls([t for t in tr if t.split()[0] not in ("bounds", "filter")], G.core(want["calls"]), 1e-6, "oracle")
    assert np.max(np.abs(k.ritzv[:nev] - np.array(want["lam"]))) < 1e-9
    assert np.max(want["res"]) < 1e-10


def _no_lanczos(lines):
    # approximate mode calls the single-vector overload ("Lanczos1 m" at kernel level) while the driver-level traces print
    # the configured "Lanczos m numvec" line: compare everything else
    return [l for l in lines if not l.startswith("Lanczos")]


def test_sequence_of_two_problems_in_approximate_mode(harness):
    """random start, then the diagonally perturbed matrix solved from the previous vectors (mode 'A': initVecs(false), no
    start-vector QR, single-vector Lanczos, caller-supplied Ritz values) - the reference driver's trace of both solves"""
    name, (N, nev, nex, deg, opt, perturb) = G.SEQ_CASE
    want = G.load(name)
    out = subprocess.run([harness, str(N), str(nev), str(nex), str(deg), str(opt), repr(perturb), "1"], check=True,
                         capture_output=True, text=True, timeout=900).stdout.splitlines()
    got = G.parse_run(out)
    assert got["iterations"] == want["iterations"] and got["filtered_vecs"] == want["filtered_vecs"]
    G.assert_same_calls(got["calls"], want["calls"], 1e-12, "own C++ driver, two-problem sequence")
    assert "initVecs 0" in want["calls"] and any(c.startswith("Lanczos1 ") for c in want["calls"])
    # oracle: the same two solves.  Tolerance 1e-4 on the scalars: the second solve's upper bound comes from a Lanczos run
    # started on a converged eigenvector (an ill-conditioned recurrence, different in every arithmetic), and the QR condition
    # estimate rho^deg amplifies it; the call sequence, widths, offsets and counts must still be identical
    H = O.clement(N, False, perturb=perturb)
    k = O.OracleCPU(H, nev, nex)
    k.config.deg, k.config.opt = deg, bool(opt)
    tr = []
    so1 = O.solve(k, tr)
    idx = np.arange(N)
    k.H[idx, idx] += 1e-3 * (idx % 7)
    k.config.approx = True
    so2 = O.solve(k, tr)
    assert so1["iterations"] + so2["iterations"] == want["iterations"]
    assert so1["filtered_vecs"] + so2["filtered_vecs"] == want["filtered_vecs"]
    G.assert_same_calls(_no_lanczos([t for t in tr if t.split()[0] not in ("bounds", "filter")]),
                        _no_lanczos(G.core(want["calls"])), 1e-4, "oracle, two-problem sequence")
    assert np.max(np.abs(k.ritzv[:nev] - np.array(want["lam"]))) < 1e-9


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_committed_traces_are_what_the_reference_driver_produces(tmp_path):
    exe = tmp_path / "ref_driver_trace"
    subprocess.run(["g++", "-std=c++17", "-O2", f"-I{REF}", "-o", str(exe),
                    os.path.join(ROOT, "tests", "golden", "ref_driver_trace.cpp")], check=True)
    for name in ("clement256", "clement512"):
        N, nev, nex, deg, opt, perturb = G.CASES[name]
        out = subprocess.run([str(exe), str(N), str(nev), str(nex), str(deg), str(opt), repr(perturb)], check=True,
                             capture_output=True, text=True, timeout=600).stdout.splitlines()
        got, want = G.parse_run(out), G.load(name)
        assert got["calls"] == want["calls"] and got["iterations"] == want["iterations"]


def test_oracle_ref_binary_reproduces_the_committed_traces():
    """oracle/_ref/ref_driver_trace, built by __graft_entry__.build() / oracle/Makefile from the reference sources"""
    if G.run_reference_driver(*G.CASES["clement256"]) is None:
        pytest.skip("oracle/_ref not built")
    for name in ("clement256", "clement256_fix", "clement512"):
        got, want = G.run_reference_driver(*G.CASES[name]), G.load(name)
        assert got["calls"] == want["calls"] and got["lam"] == want["lam"]
    name, case = G.SEQ_CASE
    assert G.run_reference_driver(*case, seq=1)["calls"] == G.load(name)["calls"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_impl_classes_instantiate_on_the_reference_interface(tmp_path):
    """ChaseHip / pChaseHip / ChaseHipPseudo / pChaseHipPseudo<T, chase::ChaseBase<T>, chase::ChaseConfig<T>> against the
    reference's own algorithm/interface.hpp:46-434: they must compile and override every pure virtual."""
    src = tmp_path / "ref_interface_check.cpp"
    src.write_text('''
#include "algorithm/algorithm.hpp"
#include "chase_amd/host/chase_hip_impl.hpp"
#include "chase_amd/host/pchase_hip_impl.hpp"
#include "chase_amd/host/chase_hip_pseudo_impl.hpp"
#include "chase_amd/host/pchase_hip_pseudo_impl.hpp"
#include <type_traits>
template <class T> using B = chase::ChaseBase<T>;
template <class T> using C = chase::ChaseConfig<T>;
using z = std::complex<double>;
#define CHECK(K, T) \\
    static_assert(std::is_base_of<B<T>, chase_amd::K<T, B<T>, C<T>>>::value, #K " derives from chase::ChaseBase"); \\
    static_assert(!std::is_abstract<chase_amd::K<T, B<T>, C<T>>>::value, #K " overrides every pure virtual")
CHECK(ChaseHip, double); CHECK(ChaseHip, z);
CHECK(pChaseHip, double); CHECK(pChaseHip, z);
CHECK(ChaseHipPseudo, double); CHECK(ChaseHipPseudo, z);
CHECK(pChaseHipPseudo, double); CHECK(pChaseHipPseudo, z);
// the reference's driver accepts them
void drive(chase_amd::ChaseHip<double, B<double>, C<double>>* k) { chase::Solve<double>(k); }
void drive(chase_amd::pChaseHipPseudo<z, B<z>, C<z>>* k) { chase::Solve_pseudo<z>(k); }
int main() { return 0; }
''')
    # ... in BOTH configurations of the reference: with -DCHASE_OUTPUT ChaseBase<T> gains the pure virtual Output()
    # (algorithm/interface.hpp:419-432), which chase_amd/host/output_override.hpp supplies (round-5 verdict: the Impls were abstract there)
    for flags in ([], ["-DCHASE_OUTPUT"]):
        p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", *flags, f"-I{REF}", f"-I{ROOT}", str(src)],
                           capture_output=True, text=True)
        assert p.returncode == 0, (flags, p.stderr[-3000:])


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_reference_driver_built_with_chase_output_runs_on_the_override_layer(tmp_path):
    """The reference's driver compiled with -DCHASE_OUTPUT (it then calls kernel->Output(...) all along the solve,
    algorithm.inc:440-2158) around the mock kernel deriving from WithOutput<chase::ChaseBase<double>> - the layer the four Impls
    derive from: it must compile (nothing left abstract), issue the committed call trace, and the messages must come out through
    the reference's logger under its own filters (algorithm/logger.hpp:156-168)."""
    exe = tmp_path / "ref_driver_trace_output"
    subprocess.run(["g++", "-std=c++17", "-O2", "-DCHASE_OUTPUT", f"-I{REF}", "-o", str(exe),
                    os.path.join(ROOT, "tests", "golden", "ref_driver_trace.cpp")], check=True)
    N, nev, nex, deg, opt, perturb = G.CASES["clement256"]
    args = [str(exe), str(N), str(nev), str(nex), str(deg), str(opt), repr(perturb)]
    want = G.load("clement256")
    texts = {}
    for level in ("error", "trace"):
        out = subprocess.run(args, check=True, capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, CHASE_LOG_LEVEL=level)).stdout
        got = G.parse_run(out.splitlines())
        assert got["calls"] == want["calls"] and got["iterations"] == want["iterations"] and got["lam"] == want["lam"]
        texts[level] = [l for l in out.splitlines() if l.split() and l.split()[0] not in
                        ("call", "lambda", "iterations", "filtered_vecs")]
    assert not texts["error"], texts["error"][:5]                      # nothing is logged at Error level on a clean solve
    assert len(texts["trace"]) > want["iterations"], texts["trace"][:5]   # the driver's per-iteration messages arrived
    # a rank filter that is not this kernel's rank silences it (the logger's rank rule reaches get_rank() through the override)
    out = subprocess.run(args, check=True, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, CHASE_LOG_LEVEL="trace", CHASE_LOG_RANK="3")).stdout
    assert [l for l in out.splitlines() if l.split() and l.split()[0] not in ("call", "lambda", "iterations", "filtered_vecs")] == []
