#!/bin/bash
# rocprofv3 kernel statistics of the default bench.py run (cfg4, 3 warm-up + 6 timed iterations = one complete solve, plus the
# full-width 4M / 3M probes).
# usage: scripts/prof_bench.sh <out_prefix under gpurun_out/>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_bench -o bench -- python3 $REPO/bench.py --no-cpu-baseline ${BENCH_ARGS:-} > ${OUT}_bench.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
python3 - "$f" > ${OUT}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (default: cfg4, one complete solve = 9 iterations, 6 of them timed; then 2 x 6 full-width probe HEMMs)")
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>11s} {'avg_us':>11s} {'pct':>7s}")
for r in rows:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:11.3f} {float(r['AverageNs'])/1e3:11.2f} {float(r['Percentage']):7.2f}")
PY
grep '"metric"' ${OUT}_bench.log | tail -1 > ${OUT}_bench.json
head -12 ${OUT}_kernel_stats.txt
