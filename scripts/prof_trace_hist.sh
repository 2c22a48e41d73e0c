#!/bin/bash
# duration histogram of every kernel of one bench solve (rocprofv3 --kernel-trace), development aid
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace -f csv -d /tmp/prof_tr -o tr -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/prof_tr.log 2>&1
f=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# keep the second half (the timed solve)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
half = rows[len(rows)//2:]
acc = collections.defaultdict(list)
for r in half:
    acc[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v.sort()
    b = [sum(1 for x in v if lo <= x < hi) for lo, hi in ((0, 20), (20, 100), (100, 500), (500, 2000), (2000, 1e9))]
    t = [sum(x for x in v if lo <= x < hi) / 1e3 for lo, hi in ((0, 20), (20, 100), (100, 500), (500, 2000), (2000, 1e9))]
    print(f"{k:70s} n={len(v):5d} total={sum(v)/1e3:8.1f} ms  counts<20us,<100,<500,<2ms,>2ms={b}  ms={[round(x,1) for x in t]}")
print("total kernel ms (second half):", tot / 1e3)
PY
