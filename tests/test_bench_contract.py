"""bench.py contract on a box without a GPU: the iteration-step timer, and the self-launching multi-rank entry point."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


class FakeSolver:
    """Stands in for the C++ driver: a 'solve' of `iters` outer iterations that calls the iteration observer."""

    def __init__(self, iters=9, nev=10):
        self.iters, self.nev, self.hook, self.calls, self.aborted = iters, nev, None, 0, 0
        self.counters = {"it": 0}

    def set_iteration_hook(self, fn):
        self.hook = fn

    def solve(self):
        self.calls += 1
        for it in range(self.iters):
            self.counters["it"] += 1
            if self.hook and self.hook(it, 100 - it, it, self.iters - it):
                self.aborted += 1
                return {"locked": 0, "t_all": 0.0}
        return {"locked": self.nev, "t_all": 1.0}


@pytest.mark.parametrize("steps,warmup", [(6, 3), (20, 5), (1, 0), (9, 0), (2, 17)])
def test_step_timer_times_exactly_k_iterations(steps, warmup):
    s = FakeSolver()
    events = []
    timer = B.StepTimer(steps, warmup, lambda: events.append("sync"), lambda: events.append("barrier"),
                        lambda: dict(s.counters))
    complete, last = B.run_timed_solves(s, timer, s.nev, lambda: ("lam", "res"))
    assert timer.done and timer.diff("it") == steps                       # exactly K iterations between the brackets
    assert timer.c0["it"] == warmup                                        # after exactly W untimed ones
    assert len(timer.per_iter) == steps
    # one bracket (sync, then barrier) on each side; the extra sync in front of each drains the solve's own queued device work so
    # that it is not booked as a bench wait (round-5 advisor)
    assert events == ["sync", "sync", "barrier", "sync", "sync", "barrier"]
    assert len(complete) >= 1 and last == ("lam", "res")                   # a complete solve backs the parity guard
    # solves run back to back; the solve in flight when the timed region ends is cut short only if a complete one exists
    total = warmup + steps
    assert s.calls == max(1, -(-total // s.iters))
    assert timer.filtered_timed == sum(100 - (i % s.iters) for i in range(warmup, total))


def test_a_solves_seconds_exclude_what_its_hooks_waited_for():
    """The join of the CPU-baseline child (before_timed) and the brackets of the timed region run inside a solve's iteration
    hooks: their seconds are not the solver's (round 5: the default short run reported 18.7 s for a 2.5 s config-2 solve)."""
    import time
    s = FakeSolver()
    timer = B.StepTimer(6, 3, lambda: None, lambda: None, lambda: dict(s.counters), before_timed=lambda: time.sleep(0.25))
    complete, _ = B.run_timed_solves(s, timer, s.nev, lambda: ("lam", "res"))
    assert len(complete) == 1
    assert complete[0]["t_all_with_bench_waits"] == 1.0 and 0.70 <= complete[0]["t_all"] <= 0.7501
    # --warmup 0: the timed region opens BEFORE the first solve - that wait belongs to no solve
    s0 = FakeSolver()
    t0 = B.StepTimer(9, 0, lambda: None, lambda: None, lambda: dict(s0.counters), before_timed=lambda: time.sleep(1.2))
    c0, _ = B.run_timed_solves(s0, t0, s0.nev, lambda: ("lam", "res"))
    assert c0[0]["t_all"] == pytest.approx(1.0, abs=1e-2)              # (only the closing bracket's microseconds are taken off)
    # a solve whose hooks waited for nothing keeps its seconds
    s2 = FakeSolver()
    t2 = B.StepTimer(9, 9, lambda: None, lambda: None, lambda: dict(s2.counters))
    c2, _ = B.run_timed_solves(s2, t2, s2.nev, lambda: ("lam", "res"))
    assert c2[-1]["t_all"] == pytest.approx(1.0, abs=1e-3)


def test_roofline_fraction_is_executed_and_bounded():
    r = B.roofline_object(model_flops=4.0e15, exec_flops=3.0e15, filt_s=50.0, calls=100, world=1)
    assert r["frac"] == pytest.approx(60.0 / B.FP64_MFMA_PEAK_TFLOPS) and r["frac"] <= 1.0
    assert r["algorithmic"] == pytest.approx(80.0) and r["achieved"] == pytest.approx(60.0)
    assert r["executed_over_model"] == pytest.approx(0.75)


def test_multi_rank_launch_starts_its_own_ranks_and_fails_only_for_lack_of_a_device():
    """`python bench.py --gpus 2` must get through the rendezvous by itself (no launcher, no RANK in the environment); on
    a box without a GPU the ranks then stop at the device check and the parent exits non-zero without a result line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "no usable device context" in p.stderr and "no HIP device visible" in p.stderr
    assert "RANK expected" not in p.stderr and "environment variable" not in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_rank_environment_binds_one_device_per_rank():
    """A rank's runtime sees ITS device only (the reference: device = node-local rank, grid/mpiGrid2D.hpp:225-233); an existing
    visibility list is honoured entry by entry; CHASE_HIP_BIND=0 and the shared-device test transport leave the lists alone."""
    env = {}
    assert B.bind_one_device(env, 3) in ("3", str(3 % max(B.kfd_gpu_count(), 1)) if B.kfd_gpu_count() else "3")
    assert "HIP_VISIBLE_DEVICES" not in env and env["CHASE_HIP_BOUND_DEVICE"] == env["ROCR_VISIBLE_DEVICES"]
    env = {"ROCR_VISIBLE_DEVICES": "4,5,6,7"}
    assert B.bind_one_device(env, 2) == "6" and env["ROCR_VISIBLE_DEVICES"] == "6"
    env = {"HIP_VISIBLE_DEVICES": "1,3", "ROCR_VISIBLE_DEVICES": "4,5,6,7", "CUDA_VISIBLE_DEVICES": "0"}
    assert B.bind_one_device(env, 1) == "7"                      # HIP's entry 3 indexes the runtime's list
    assert "HIP_VISIBLE_DEVICES" not in env and "CUDA_VISIBLE_DEVICES" not in env
    env = {"CHASE_HIP_BIND": "0", "HIP_VISIBLE_DEVICES": "0,1"}
    assert B.bind_one_device(env, 1) is None and env == {"CHASE_HIP_BIND": "0", "HIP_VISIBLE_DEVICES": "0,1"}
    env = {"CHASE_HIP_TRANSPORT": "host"}
    assert B.bind_one_device(env, 1) is None and "ROCR_VISIBLE_DEVICES" not in env


def test_spawned_ranks_carry_the_binding(monkeypatch):
    """spawn_ranks hands child r an environment whose runtime visibility is device r - set before the child starts"""
    seen = []

    class FakePopen:
        def __init__(self, cmd, env=None, **kw):
            seen.append(env)
        def poll(self): return 0
        def terminate(self): pass
        def kill(self): pass

    monkeypatch.setattr(B.subprocess, "Popen", FakePopen)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CHASE_HIP_TRANSPORT", raising=False)
    monkeypatch.setattr(B, "kfd_gpu_count", lambda: 8)

    class A:
        gpus = 4
    rc = B.spawn_ranks(A, ["--gpus", "4"], "bound")
    assert rc == 1                                           # no result line from fake ranks
    assert [e["ROCR_VISIBLE_DEVICES"] for e in seen] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in seen] == ["0", "1", "2", "3"] and all(e["WORLD_SIZE"] == "4" for e in seen)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in seen)
    assert all(e["CHASE_BENCH_MODE"] == "bound" for e in seen)              # the ranks do not probe again
    # all devices visible, device = local rank (the reference's way, grid/mpiGrid2D.hpp:225-233)
    del seen[:]
    B.spawn_ranks(A, ["--gpus", "4"], "unbound")
    assert all("ROCR_VISIBLE_DEVICES" not in e and e["CHASE_HIP_BIND"] == "0" for e in seen)


def test_mode_selection_tries_bound_then_unbound_then_threads(monkeypatch):
    """`--ranks auto`: the way the ranks hold their devices is decided by probe children (context + RCCL communicators + the
    256 MB all-reduce proof) before the deciding process touches a GPU; the probes are answered by a test hook here.  As the
    parent of its own ranks and as one rank of a torch.distributed.run launch the decision is the same."""
    class A:
        gpus = 4
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(key, raising=False)
    for spec, want in (({"bound": 0, "unbound": 0}, "bound"), ({"bound": 5, "unbound": 0}, "unbound"),
                       ({"bound": 124, "unbound": 4}, "threads")):
        monkeypatch.setenv("CHASE_BENCH_FAKE_PROBE", json.dumps(spec))
        assert B.choose_mode_as_parent(A, ["--gpus", "4"]) == want
        monkeypatch.setenv("RANK", "2"); monkeypatch.setenv("WORLD_SIZE", "4"); monkeypatch.setenv("LOCAL_RANK", "2")
        assert B.choose_mode_as_rank(A, ["--gpus", "4"]) == want
        for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            monkeypatch.delenv(key, raising=False)


def test_thread_mode_under_a_launcher_leaves_the_gpu_to_rank_zero(tmp_path):
    """`--ranks threads` under torch.distributed.run: rank 0 hosts one thread per GPU, every other rank leaves with status 0
    before it imports anything that touches a GPU (no output, no device context)."""
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ranks", "threads", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0 and p.stdout.strip() == "" and "device" not in p.stderr


def test_cpu_baseline_runs_as_a_child_and_is_joined_before_the_timed_region():
    job = B.CpuBaselineJob(384, True, 32, 0.5)
    s = FakeSolver()
    order = []
    timer = B.StepTimer(2, 1, lambda: order.append("sync"), lambda: None, lambda: dict(s.counters),
                        before_timed=lambda: (order.append("join"), job.join()))
    B.run_timed_solves(s, timer, s.nev, lambda: ("lam", "res"))
    # (drain of the solve's queued work,) then the join, then the opening bracket's sync: the child is over before the bracket
    assert order[:3] == ["sync", "join", "sync"]
    rec = job.result()
    assert rec["value"] > 0 and rec["kind"] == "port" and "child process" in rec["ran"]
    assert rec["sample_shape"]["N"] == 384


def test_converged_flag_holds_the_solvers_own_tolerance_on_recomputed_residuals():
    import numpy as np
    lam = np.arange(4.0)
    tol = 1e-10
    ok = {"ok": True}
    assert B.converged_ok(lam, [5e-11, 9.9e-11, 3e-10, 1e-11], [5e-11, 1.0005e-10, 3e-10, 1e-11], tol, ok)   # early-locked pair above tol is fine
    assert not B.converged_ok(lam, [5e-11, 9.9e-11, 3e-10, 1e-11], [5e-11, 1.01e-10, 3e-10, 1e-11], tol, ok)  # converged by the solver, not by H
    assert not B.converged_ok(lam, [5e-11] * 4, [5e-11] * 4, tol, {"ok": False})
    assert not B.converged_ok([0, 1, np.nan, 3], [5e-11] * 4, [5e-11] * 4, tol, ok)


def test_mode_selection_under_the_real_launcher_without_a_gpu():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` (the driver's multi-GPU call) on a box without a
    GPU: every rank settles the mode through its own probe children - which must find each other on their own port although
    the launcher's agent store owns MASTER_PORT -, both process modes fail for lack of a device, the thread mode is refused for
    the same reason, and the job ends non-zero without a result line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                           "CHASE_HIP_TRANSPORT", "CHASE_BENCH_FAKE_PROBE")}
    env["CHASE_BENCH_PROBE_TIMEOUT"] = "120"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert "probe of mode 'bound': exit status 3" in p.stderr and "probe of mode 'unbound': exit status 3" in p.stderr
    assert "no usable device context" in p.stderr                          # the probes met and agreed (status 3, not a timeout)
    assert "--ranks threads needs one GPU per rank" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
