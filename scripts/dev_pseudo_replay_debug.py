"""development: the pseudo-Hermitian replay of tests/test_gpu_replay.py step by step with a watchdog that dumps all thread stacks"""
import faulthandler, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
log = open(os.path.join(ROOT, "gpurun_out", "pseudo_replay_debug.log"), "w")
faulthandler.dump_traceback_later(70, exit=True, file=log)
def say(*a):
    print(time.strftime("%H:%M:%S"), *a, file=log, flush=True)
import numpy as np
from rank_threads import run_ranks
from chase_amd import dist as cd
from chase_amd.capi import Context, tape_mode, tape_get, tape_load
import test_gpu_replay as T
nprow, npcol = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2x2").split("x"))
N, nev, nex = 1024, 24, 16
real = {}
say("real ranks start")
run_ranks(nprow, npcol, T._real_rank_pseudo, N, nev, nex, real)
say("real ranks done", real[0]["stats"]["iterations"], real[0]["stats"]["filtered_vecs"], real[0]["stats"]["locked"], real[0]["resid"], len(real[0]["tape"]))
ctx = Context(0)
rl, cl = cd.Layout(N, 0, nprow), cd.Layout(N, 0, npcol)
for r in range(nprow * npcol):
    g = cd.Grid(ctx, nprow, npcol, r, transport="loopback")
    dH = cd.gen_bse_local(ctx, N, True, rl, cl, g.myrow, g.mycol, dmin=1.0, dmax=11.0, offdiag=1e-3)
    s = cd.DistPseudoSolver(ctx, g, dH, N, nev, nex, True, 0, 0)
    s.set(device_rng=1, numlanczos=10, lanczositer=50)
    tape_load(s, real[r]["tape"]); tape_mode(s, 2)
    s.set_iteration_hook(lambda it, f, l, u: say("  replay rank", r, "iteration", it, f, l, u) or False)
    say("replay rank", r, "start")
    ctx.oplog(True)
    import threading
    done = threading.Event()
    def watchdog(rr=r):
        if not done.wait(25.0):
            lines = ctx.oplog_lines()
            say("WATCHDOG: rank", rr, "still solving after 25 s; oplog has", len(lines), "lines; the last 40:")
            for l in lines[-40:]:
                say("   ", l)
    threading.Thread(target=watchdog, daemon=True).start()
    st = s.solve(trace=True)
    done.set()
    ctx.oplog(False)
    log_lines = ctx.oplog_lines()
    say("replay rank", r, "done", st["iterations"], st["filtered_vecs"], "tolerated", s.get("tape_tolerated"), "qr retries", s.get("tape_qr_retries"),
        "oplog", len(log_lines), len(real[r]["oplog"]), "equal", log_lines == real[r]["oplog"])
    if log_lines != real[r]["oplog"]:
        for i, (a, b) in enumerate(zip(log_lines, real[r]["oplog"])):
            if a != b:
                say("   first difference at", i, a, "|", b); break
    s.close(); dH.free(); g.close()
say("all done")
