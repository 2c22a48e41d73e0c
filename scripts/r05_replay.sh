#!/bin/bash
# Round 5: single-rank replay of the config-4 solve (bench.py --replay-rank) + rocprofv3 kernel statistics of the 4x2 replay.
# usage: scripts/r05_replay.sh <tag>   -> gpurun_out/<tag>_replay.json, <tag>_replay_4x2_kernel_stats.txt, ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05}
OUT=$REPO/gpurun_out/$TAG
GRIDS=${GRIDS:-4x2,2x2,2x1}
WL=${WL:-cfg4}
TAPE=${TAPE:-$REPO/gpurun_out/${TAG}_${WL}_tape.npz}
case $TAPE in /*) ;; *) TAPE=$REPO/$TAPE ;; esac          # (the profiled runs start from /tmp)
mkdir -p $REPO/gpurun_out
cd $REPO
if [ "${SKIP_MAIN:-0}" != "1" ]; then
python3 bench.py --replay-rank $GRIDS --workload $WL --tape $TAPE --oplog-out ${OUT}_oplog_%g.txt > ${OUT}_replay.json 2> ${OUT}_replay.log || { tail -20 ${OUT}_replay.log; exit 1; }
tail -4 ${OUT}_replay.log
fi
cd /tmp && export TMPDIR=/tmp
for PROF_GRID in ${PROF_GRIDS:-4x2 2x2 2x1}; do
rm -rf /tmp/prof_replay
CHASE_HIP_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats -f csv -d /tmp/prof_replay -o replay -- python3 $REPO/bench.py --replay-rank ${PROF_GRID:-4x2} --workload $WL --tape $TAPE > ${OUT}_replay_prof_${PROF_GRID}.json 2> ${OUT}_replay_prof_${PROF_GRID}.log || { tail -20 ${OUT}_replay_prof_${PROF_GRID}.log; exit 1; }
f=$(find /tmp/prof_replay -name "*kernel_stats.csv" | head -1)
python3 - "$f" "${PROF_GRID:-4x2}" "$WL" > ${OUT}_replay_${PROF_GRID:-4x2}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f"rocprofv3 --kernel-trace --marker-trace --stats -- python3 bench.py --replay-rank {sys.argv[2]} --workload {sys.argv[3]} --tape <recorded>   (CHASE_HIP_ROCTX=1: one rank of the grid, loopback transport, the taped call sequence of the real single-GPU solve)")
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>11s} {'avg_us':>11s} {'pct':>7s}")
for r in rows:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:11.3f} {float(r['AverageNs'])/1e3:11.2f} {float(r['Percentage']):7.2f}")
PY
kt=$(find /tmp/prof_replay -name "*kernel_trace.csv" | head -1)
mt=$(find /tmp/prof_replay -name "*marker_api_trace.csv" | head -1)
ls -la /tmp/prof_replay/* > ${OUT}_replay_prof_files.txt 2>&1
head -3 "$kt" > ${OUT}_kernel_trace_head.txt 2>/dev/null
head -5 "$mt" > ${OUT}_marker_trace_head.txt 2>/dev/null
python3 $REPO/scripts/phase_table.py "$kt" "$mt" > ${OUT}_replay_${PROF_GRID:-4x2}_phase_table.txt 2>&1 || true
head -8 ${OUT}_replay_${PROF_GRID:-4x2}_kernel_stats.txt
done
