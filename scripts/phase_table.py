#!/usr/bin/env python3
"""Per-phase kernel table from `CHASE_HIP_ROCTX=1 rocprofv3 --kernel-trace --marker-trace -f csv` output: every kernel of the
kernel trace is attributed to the innermost chase:* range (chase_amd/host/roctx.hpp: a range ends only after its phase's device
work is complete) whose [start, end] contains the kernel's start, and the table lists per phase its total device time and its
top kernels.  usage: phase_table.py <kernel_trace.csv> <marker_api_trace.csv>"""
import bisect
import csv
import sys
from collections import defaultdict


def col(row, *names):
    for n in names:
        if n in row:
            return row[n]
    raise KeyError(names)


def main(kt, mt):
    ranges = []
    for r in csv.DictReader(open(mt)):
        name = col(r, "Function", "Name", "Message")
        if not name.startswith("chase:"):
            continue
        ranges.append((int(col(r, "Start_Timestamp", "Start")), int(col(r, "End_Timestamp", "End")), name[6:]))
    ranges.sort()
    starts = [a for a, _, _ in ranges]
    per = defaultdict(lambda: defaultdict(lambda: [0, 0]))
    tot = defaultdict(float)
    for r in csv.DictReader(open(kt)):
        s, e = int(col(r, "Start_Timestamp", "Start")), int(col(r, "End_Timestamp", "End"))
        name = col(r, "Kernel_Name", "Name")
        i = bisect.bisect_right(starts, s) - 1
        phase = "(outside)"
        # innermost enclosing range: walk back over the candidates that started before the kernel
        j = i
        while j >= 0:
            a, b, nm = ranges[j]
            if a <= s <= b:
                phase = nm
                break
            j -= 1
            if i - j > 64:
                break
        per[phase][name][0] += 1
        per[phase][name][1] += e - s
        tot[phase] += e - s
    calls = defaultdict(int)
    for a, b, nm in ranges:
        calls[nm] += 1
    wall = {nm: sum(b - a for a, b, n2 in ranges if n2 == nm) for nm in calls}
    grand = sum(tot.values())
    print(f"{'phase':22s} {'ranges':>7s} {'range_s':>9s} {'kernel_s':>9s} {'share':>7s}")
    for ph in sorted(tot, key=lambda p: -tot[p]):
        print(f"{ph:22s} {calls.get(ph, 0):7d} {wall.get(ph, 0) / 1e9:9.3f} {tot[ph] / 1e9:9.3f} {100 * tot[ph] / grand:6.2f}%")
    for ph in sorted(tot, key=lambda p: -tot[p]):
        print(f"\n== {ph}: {tot[ph] / 1e9:.3f} s of kernels")
        for name, (n, ns) in sorted(per[ph].items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"   {name[:100]:100s} {n:7d} {ns / 1e6:11.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
