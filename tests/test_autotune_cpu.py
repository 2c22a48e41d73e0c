"""The selection logic of the first-contact self-tuning (chase_amd/autotune.py) on scripted timings - no GPU: the schedule is
one factor at a time, the incumbent stays unless a candidate is faster by more than the tie margin (or as fast with less
exposed communication), and the trial budget cuts the schedule from the end."""
from chase_amd import autotune as A

BASE = {"panel_cols": 256, "panel_rounds": 4, "comm_streams": 2}


def script(times):
    """times: {(panel, rounds, streams): (seconds, exposed_ms)}; unknown settings are slow"""
    calls = []

    def measure(s):
        key = (s["panel_cols"], s["panel_rounds"], s["comm_streams"])
        calls.append(key)
        sec, exp = times.get(key, (9.0, 9.0))
        return {"seconds": sec, "exposed_ms": exp}
    return measure, calls


def test_schedule_is_one_factor_at_a_time_from_the_incumbent():
    m, calls = script({(256, 4, 2): (1.00, 5.0), (128, 4, 2): (1.10, 3.0), (512, 4, 2): (0.90, 6.0),
                       (512, 0, 2): (0.95, 6.0), (512, 4, 1): (0.80, 9.0)})
    best, table = A.tune(BASE, 5, m)
    assert calls == [(256, 4, 2), (128, 4, 2), (512, 4, 2), (512, 0, 2), (512, 4, 1)]      # later stages start from the winner
    assert best == {"panel_cols": 512, "panel_rounds": 4, "comm_streams": 1}
    assert [r["kept"] for r in table] == [False, False, False, False, True]
    assert len(table) == 5 and table[0]["setting"] == BASE


def test_incumbent_stays_inside_the_tie_margin_unless_less_is_exposed():
    m, _ = script({(256, 4, 2): (1.000, 5.0), (128, 4, 2): (0.995, 5.0), (512, 4, 2): (1.004, 2.0),
                   (512, 0, 2): (1.2, 0.0), (512, 4, 1): (1.003, 2.0)})
    best, table = A.tune(BASE, 5, m)
    # 128 is 0.5 % faster with the same exposed time: a tie, the incumbent stays; 512 is as fast and exposes less: taken;
    # one stream is as fast and exposes as much: the incumbent (two streams) stays
    assert best == {"panel_cols": 512, "panel_rounds": 4, "comm_streams": 2}
    assert [r["kept"] for r in table] == [False, False, True, False, False]


def test_budget_cuts_the_schedule_from_the_end():
    m, calls = script({(256, 4, 2): (1.0, 1.0)})
    best, table = A.tune(BASE, 1, m)
    assert best == BASE and calls == [(256, 4, 2)] and table[0]["kept"]
    m, calls = script({(256, 4, 2): (1.0, 1.0), (128, 4, 2): (0.5, 1.0)})
    best, _ = A.tune(BASE, 3, m)
    assert calls == [(256, 4, 2), (128, 4, 2), (512, 4, 2)] and best["panel_cols"] == 128
    assert A.plan(BASE, 5) == [("panel_cols", [128, 512]), ("panel_rounds", [0]), ("comm_streams", [1])]
    one = {"panel_cols": 512, "panel_rounds": 0, "comm_streams": 1}
    assert A.plan(one, 9, streams=True) == [("panel_cols", [128, 256]), ("panel_rounds", [4]), ("comm_streams", [2])]
    # round-5 advisor: TWO communication streams are tried only when asked for (CHASE_HIP_AUTOTUNE_STREAMS=1; never seen on
    # xGMI); what the schedule leaves out - and why - is recorded for the bench's JSON line
    skipped = []
    assert A.plan(one, 9, streams=False, skipped=skipped) == [("panel_cols", [128, 256]), ("panel_rounds", [4])]
    assert skipped == [{"knob": "comm_streams", "value": 2, "why": "opt-in: CHASE_HIP_AUTOTUNE_STREAMS=1 (never run on xGMI)"}]
    skipped = []
    assert A.plan(one, 2, streams=True, skipped=skipped) == [("panel_cols", [128])]
    assert [(k["knob"], k["value"]) for k in skipped] == [("panel_cols", 256), ("panel_rounds", 4), ("comm_streams", 2)]


def test_a_slower_candidate_never_replaces_a_faster_incumbent_whatever_it_exposes():
    assert not A.better({"seconds": 1.02, "exposed_ms": 0.0}, {"seconds": 1.0, "exposed_ms": 50.0})
    assert A.better({"seconds": 0.98, "exposed_ms": 80.0}, {"seconds": 1.0, "exposed_ms": 5.0})
    assert A.better({"seconds": 1.0, "exposed_ms": 1.0}, {"seconds": 1.0, "exposed_ms": 5.0})
    assert not A.better({"seconds": 1.0, "exposed_ms": 5.0}, {"seconds": 1.0, "exposed_ms": 5.0})


def test_small_collective_latency_vetoes_a_setting_that_only_speeds_up_the_filter():
    """What RCCL's socket transport showed (profiles/r05_socket_rccl_streams.txt): two communication streams make the filter
    products 10 % faster and every small synchronous collective 19 ms instead of 1 ms - a solve issues ~2.8 of those per
    product, so the setting must lose; with xGMI-like latencies (60 vs 50 us) the same filter gain must win."""
    def script2(lat2_us):
        def measure(s):
            if s["comm_streams"] == 2:
                return {"seconds": 0.0527, "exposed_ms": 39.7, "small_collective_us": lat2_us}
            return {"seconds": 0.0588, "exposed_ms": 41.7, "small_collective_us": 950.0 if lat2_us > 1000 else 50.0}
        return measure
    base = {"panel_cols": 256, "panel_rounds": 4, "comm_streams": 1}
    best, table = A.tune(base, 5, script2(19000.0), streams=True)
    assert best["comm_streams"] == 1 and table[-1]["setting"]["comm_streams"] == 2 and not table[-1]["kept"]
    best, table = A.tune(base, 5, script2(60.0), streams=True)
    assert best["comm_streams"] == 2 and table[-1]["kept"]
    best, table = A.tune(base, 5, script2(60.0), streams=False)                 # the default: not even tried
    assert best["comm_streams"] == 1 and all(r["setting"]["comm_streams"] == 1 for r in table)
    assert A.cost({"seconds": 0.05, "exposed_ms": 0, "small_collective_us": 1000.0}) == pytest.approx(0.05 + A.SMALL_PER_PRODUCT * 1e-3)


import pytest  # noqa: E402
