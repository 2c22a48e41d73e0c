"""Reader / comparer of tests/golden/driver_trace_*.txt — runs of the REFERENCE's own driver (chase::Solve compiled from
/root/reference by tests/golden/make_driver_traces.sh) on the naive CPU kernel of tests/cpu_mock_kernel.hpp."""
import os
import re

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# name -> (N, nev, nex, deg, opt, perturb): the command lines of make_driver_traces.sh
CASES = {
    "clement256": (256, 24, 16, 16, 1, 1e-6),
    "clement256_fix": (256, 24, 16, 20, 0, 0.0),
    "clement512": (512, 50, 14, 10, 1, 1e-6),
    "clement1001": (1001, 60, 40, 20, 1, 1e-6),
    "clement1200": (1200, 80, 60, 20, 1, 1e-6),
}
# two-problem sequence: the first solve from random vectors, then H[i,i] += 1e-3 * (i mod 7) and a solve in approximate mode
# from the first solve's vectors (single-vector Lanczos for the upper bound, no start-vector QR)
SEQ_CASE = ("clement256_seq", (256, 24, 16, 16, 1, 1e-6))
# the calls the product's driver-side trace (chase_hip_solver_trace) and the oracle's trace record as well
CORE = ("initVecs", "QR", "Lanczos", "HEMM", "RR", "Resd", "Lock")
_NUM = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$")


def parse_run(lines):
    out = {"lam": [], "res": [], "calls": [], "trace": []}
    for l in lines:
        t = l.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] in ("iterations", "filtered_vecs", "locked", "stats_iterations", "stats_filtered_vecs"):
            out[t[0]] = int(t[1])
        elif t[0] == "lambda":
            out["lam"].append(float(t[1])); out["res"].append(float(t[2]))
        elif t[0] == "call":
            out["calls"].append(l.split(" ", 1)[1].strip())
        elif t[0] == "trace":
            out["trace"].append(l.split(" ", 1)[1].strip())
    return out


def load(name):
    with open(os.path.join(GOLDEN, f"driver_trace_{name}.txt")) as f:
        return parse_run(f.read().splitlines())


def core(calls):
    return [c for c in calls if c.split()[0] in CORE]


def same_line(a, b, rtol):
    ta, tb = a.split(), b.split()
    if len(ta) != len(tb):
        return False
    for x, y in zip(ta, tb):
        if _NUM.match(x) and _NUM.match(y):
            fx, fy = float(x), float(y)
            if abs(fx - fy) > rtol * max(1.0, abs(fx), abs(fy)):
                return False
        elif x != y:
            return False
    return True


def assert_same_calls(got, want, rtol, what):
    assert len(got) == len(want), f"{what}: {len(got)} calls, the reference driver issued {len(want)}"
    for i, (a, b) in enumerate(zip(got, want)):
        assert same_line(a, b, rtol), f"{what}: call {i}: got '{a}', the reference driver issued '{b}'"


REF_BIN = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "ref_driver_trace")


def run_reference_driver(N, nev, nex, deg, opt, perturb, seq=0):
    """Runs oracle/_ref/ref_driver_trace - the reference's own chase::Solve built from the reference sources by
    oracle/Makefile (it travels to the GPU box with the snapshot) - and parses its output; None if it was not built."""
    import subprocess
    exe = os.path.normpath(REF_BIN)
    if not os.path.exists(exe):
        return None
    out = subprocess.run([exe, str(N), str(nev), str(nex), str(deg), str(opt), repr(perturb), str(seq)], check=True,
                         capture_output=True, text=True, timeout=900).stdout.splitlines()
    return parse_run(out)
