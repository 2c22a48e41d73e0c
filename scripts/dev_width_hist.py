"""Where the filter's time goes by active width: one traced solve of a bench workload, then every distinct HEMM width of its
filter calls is timed on its own (filter-phase kernel symbol, beta != 0) and weighted with its number of calls.
usage: dev_width_hist.py cfg3 [reps]"""
import os
import sys
from collections import Counter
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from chase_amd.capi import Context, Solver, lib, check

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, cplx, nev, nex = B.WORKLOADS[wl]
F = 4 if cplx else 1
with Context(0) as ctx:
    dH = ctx.gen_clement(N, cplx, scale=B.MATRIX_SCALE / N, perturb=B.MATRIX_PERTURB, seed=42)
    s = Solver(ctx, None, nev, nex, h_on_device_ptr=dH.ptr, N=N, cplx=cplx)
    s.set(device_rng=1)
    st = s.solve(trace=True)
    widths = Counter()
    for l in s.trace():
        t = l.split()
        if len(t) >= 2 and t[0] == "HEMM":
            widths[int(t[1])] += 1
    filt_ms = s.get("filter_ms")
    print(f"{wl}: {st['iterations']} iterations, {st['filtered_vecs']} vectors, filter {filt_ms:.1f} ms "
          f"= {2.0 * F * N * N * st['filtered_vecs'] / filt_ms / 1e9:.2f} TFLOP/s (model), {sum(widths.values())} HEMM calls "
          f"{len(widths)} distinct widths")
    s.close()
    dt = np.complex128 if cplx else np.float64
    n_max = max(widths)
    dB = ctx.empty((N, n_max), dt); dC = ctx.empty((N, n_max), dt)
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), N, n_max, dB.ptr, N, 0, 0, N, 2), "fill")
    check(lib.chase_hip_fill_normal(ctx.h, int(cplx), N, n_max, dC.ptr, N, 0, 0, N, 3), "fill")
    lib.chase_hip_ctx_set_phase(ctx.h, 1)
    rows = []
    for w in sorted(widths):
        ctx.gemm("N", N, w, N, 0.5, dH.ptr, N, dB.ptr, N, 0.25, dC.ptr, N, cplx)
        ctx.timer_start()
        for _ in range(reps):
            ctx.gemm("N", N, w, N, 0.5, dH.ptr, N, dB.ptr, N, 0.25, dC.ptr, N, cplx)
        ms = ctx.timer_stop() / reps
        rows.append((w, widths[w], ms))
    tot = sum(c * ms for _, c, ms in rows)
    print(f"sum over calls of the per-width launch time: {tot:.1f} ms")
    edges = [0, 16, 32, 64, 128, 256, 512, 1024, 1 << 30]
    print(" width bin      calls     vectors   time ms   share   TFLOP/s(model)")
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = [(w, c, ms) for w, c, ms in rows if lo < w <= hi]
        if not sel:
            continue
        t = sum(c * ms for _, c, ms in sel); v = sum(c * w for w, c, _ in sel)
        print(f" {lo + 1:5d}-{min(hi, n_max):5d} {sum(c for _, c, _ in sel):8d} {v:10d} {t:9.1f} {100 * t / tot:6.1f}% {2.0 * F * N * N * v / t / 1e9:8.2f}")
    if os.environ.get("DEV_WIDTHS_ALL"):
        for w, c, ms in rows:
            print(f"  w={w:5d} calls={c:4d} {ms:8.3f} ms {2.0 * F * N * N * w / ms / 1e9:7.2f} TFLOP/s")
