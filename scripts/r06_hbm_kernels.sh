#!/bin/bash
# HBM-roofline table of the streaming kernels (round-5 verdict, item 6): un-profiled timings, then one rocprofv3 pass per TCC
# counter (FETCH_SIZE and WRITE_SIZE cannot share a pass) over the same launches.  usage: scripts/r06_hbm_kernels.sh <outdir>
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $REPO/scripts/r06_hbm_kernels.py 65536 2560 5 $OUT/hbm_kernels.json > $OUT/hbm_kernels_timed.txt 2>&1 || { tail -5 $OUT/hbm_kernels_timed.txt; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/hbm_pmc_$c; rm -rf $d
  rocprofv3 --kernel-trace --pmc $c -f csv -d $d -- python3 $REPO/scripts/r06_hbm_kernels.py 65536 2560 1 > /tmp/hbm_pmc_$c.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c > $OUT/hbm_kernels_$c.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{sys.argv[2]:12s} launches={len(v):3d} mean_KiB={sum(v)/len(v):.6g} last_KiB={v[-1]:.6g} kernel={k}")
PY
done
# per-kernel durations of an un-counted profiled run
d=/tmp/hbm_stats; rm -rf $d
rocprofv3 --kernel-trace --stats -f csv -d $d -- python3 $REPO/scripts/r06_hbm_kernels.py 65536 2560 3 > /tmp/hbm_stats.log 2>&1
f=$(find $d -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/hbm_kernels_kernel_stats.csv
