"""The scalar tape (chase_amd/host/tape.hpp) without a GPU: the product's driver on the CPU mock kernel, recorded on one
problem and replayed on a kernel that holds ANOTHER matrix - the replayed driver must issue the recording's virtual-call
sequence call for call (that is what makes a single rank of a multi-GPU solve measurable on its own, bench.py --replay-rank),
although the second kernel's own numbers would have steered it elsewhere (the control run)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path_factory.mktemp("tape") / "tape_harness"
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", str(exe), os.path.join(ROOT, "tests", "tape_harness.cpp")],
                   check=True, cwd=ROOT)
    return str(exe)


@pytest.mark.parametrize("N,nev,nex,deg", [(96, 8, 6, 10), (120, 12, 8, 20)])
def test_replay_issues_the_recorded_call_sequence_on_another_kernel(harness, N, nev, nex, deg):
    out = subprocess.run([harness, str(N), str(nev), str(nex), str(deg)], check=True, capture_output=True, text=True,
                         timeout=600).stdout.splitlines()
    calls = {k: [l[2:] for l in out if l.startswith(k + " ") and not l.split()[1] in ("iterations",)] for k in "ABC"}
    head = {k: next(l for l in out if l.startswith(k + " iterations")) .split() for k in "ABC"}
    assert len(calls["A"]) > 50
    # the replayed driver on kernel B: A's calls, every scalar argument included (they are printed with 17 digits)
    assert calls["B"] == calls["A"]
    assert head["B"][2] == head["A"][2] and head["B"][4] == head["A"][4] and head["B"][6] == head["A"][6]
    # the whole tape was consumed, and the driver was shown A's Ritz values
    size = int(next(l for l in out if l.startswith("tape_size")).split()[1])
    assert int(head["B"][8]) == size
    for l in out:
        if l.startswith("ritz "):
            a, b = l.split()[1:]
            assert a == b
    # the control: kernel B on its own numbers takes another path
    assert calls["C"] != calls["A"]
    # a tape that ends early stops the replay instead of letting it run on
    assert next(l for l in out if l.startswith("truncated")).startswith("truncated tape:")


def test_replay_of_the_pseudo_hermitian_driver(harness):
    """chase::Solve_pseudo through the tape (round 5: bench.py --replay-rank on the pseudo-Hermitian workload): a scripted kernel
    A is recorded, another script B replayed - B's driver issues A's calls (H^2 filter steps, K-conjugations, the +- Ritz pairs of
    rayleighRitz_v2 taken from the tape: 2 * block values per RR frame)"""
    out = subprocess.run([harness, "pseudo"], check=True, capture_output=True, text=True, timeout=600).stdout.splitlines()
    calls = {k: [l[2:] for l in out if l.startswith(k + " ") and not l.split()[1] in ("iterations",)] for k in "ABC"}
    head = {k: next(l for l in out if l.startswith(k + " iterations")).split() for k in "ABC"}
    assert len(calls["A"]) > 50 and any(c.startswith("HEMM_H2") for c in calls["A"]) and any(c.startswith("ApplyKconjugate") for c in calls["A"])
    assert calls["B"] == calls["A"]
    assert head["B"][2] == head["A"][2] and head["B"][4] == head["A"][4] and head["B"][6] == head["A"][6]
    assert int(head["B"][8]) == int(next(l for l in out if l.startswith("tape_size")).split()[1])
    ritz = [l.split()[1:] for l in out if l.startswith("ritz ")]
    assert len(ritz) == 16 and all(a == b for a, b in ritz)
    assert calls["C"] != calls["A"]
    assert next(l for l in out if l.startswith("truncated")).startswith("truncated tape:")
