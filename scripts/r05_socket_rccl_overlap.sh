#!/bin/bash
# Round 5: the overlap machinery against REAL asynchronous RCCL collectives (config 2 on 2x2, four rank processes sharing the one
# GPU, one NCCL_HOSTID per rank -> socket transport, ~5 GB/s): panel pipeline on / off, one / two communication streams.
# Functional evidence that the per-panel events order the streams and that waits shrink; says nothing about xGMI.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r05s}
cd $REPO
for v in "pipeline1_streams2 CHASE_HIP_PIPELINE=1 CHASE_HIP_COMM_STREAMS=2" "pipeline1_streams1 CHASE_HIP_PIPELINE=1 CHASE_HIP_COMM_STREAMS=1" "pipeline0_streams2 CHASE_HIP_PIPELINE=0 CHASE_HIP_COMM_STREAMS=2" "pipeline0_streams1 CHASE_HIP_PIPELINE=0 CHASE_HIP_COMM_STREAMS=1"; do
  set -- $v; name=$1; shift
  env CHASE_BENCH_FAKE_HOSTS=1 "$@" python3 bench.py --gpus 4 --workload cfg2 --steps 9 --warmup 0 --no-cpu-baseline --no-probe --no-autotune > ${OUT}_$name.json 2> ${OUT}_$name.log || { tail -5 ${OUT}_$name.log; exit 1; }
  python3 - ${OUT}_$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "solve %.2f s" % d["solve_seconds"], "exposed %.0f ms" % d["comm_exposed_ms"], "waits", d["comm_waits"], "iterations", d["iterations_per_solve"], "vectors", d["filtered_vecs_per_solve"], "converged", d["converged"])
PY
done
