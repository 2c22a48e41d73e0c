"""The event-ordering state machine of the panel-pipelined distributed HEMM (chase_amd/host/panel_pipeline.hpp - the code
pChaseHip::hemm_ptr runs) without a GPU: bound to a simulator of HIP streams and events that mirrors the grid's collective
logic (tests/pipeline_harness.cpp), a filter-like sequence of alternating products is checked for unordered conflicting
accesses - for every combination of active / inactive groups (4 x 2 and 2 x 2: both; 2 x 1: only the column group; 1 x 2-like:
only the row group), one and two communication streams and several panel widths.  The checker is validated by the rule round 4
first shipped (panelise only when the product's OWN group is active): with it the one-column grid must show the race that the
first real RCCL run between two ranks found."""
import itertools
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path_factory.mktemp("pipe") / "pipeline_harness"
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", str(exe), os.path.join(ROOT, "tests", "pipeline_harness.cpp")], check=True, cwd=ROOT)
    return str(exe)


def run(harness, *args):
    return subprocess.run([harness, *map(str, args)], check=True, capture_output=True, text=True, timeout=120).stdout


@pytest.mark.parametrize("col,row", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_every_conflicting_access_is_ordered(harness, col, row):
    for streams, panel, pipelined in itertools.product((1, 2), (64, 256, 512, 4096), (1, 0)):
        out = run(harness, col, row, streams, panel, "fixed", pipelined)
        assert out.startswith("OK "), (col, row, streams, panel, pipelined, out)


def test_the_simulator_finds_the_race_of_the_one_column_grid(harness):
    """2 x 1 grid (column group active, row group a single rank) with round 4's first rule: the row -> column product is issued
    as one unsynchronised GEMM while the column group's asynchronous all-reduces of its input are still in flight"""
    for streams in (1, 2):
        out = run(harness, 1, 0, streams, 256, "r4bug")
        assert out.startswith("HAZARD"), out
        assert "unordered after write by allreduce" in out
    # with both groups active the old rule was fine (which is why only the first two-rank run over real communicators found it)
    assert run(harness, 1, 1, 1, 256, "r4bug").startswith("OK ")
