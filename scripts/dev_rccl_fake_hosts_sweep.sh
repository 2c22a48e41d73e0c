#!/bin/bash
# Every scenario of tests/dist_scenarios.py that has a command-line form, over REAL RCCL communicators (one NCCL_HOSTID per rank
# process, socket transport) on 2, 3 and 4 ranks of one GPU.  One-off sweep (the driver-run subset is in tests/test_gpu_processes.py).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export LINES_SHOWN=1 FAKE_TIMEOUT=150
port=29700
run() { port=$((port+1)); PORT=$port bash $REPO/scripts/dev_rccl_fake_hosts.sh "$@" | head -1; }
for n in 2 4; do
  run $n hemm_kat
  run $n ops d 16; run $n ops z 0
  run $n reference_units d 0; run $n reference_units z 16
  run $n symcheck d 16; run $n symcheck z 0
  run $n qr_fixtures d 8; run $n qr_fixtures z 0
  run $n pseudo_ops 16; run $n pseudo_solve_real 0; run $n pseudo_solve_real 16; run $n pseudo_solve 32
  run $n solve 256 24 16 z 0 16; run $n solve 1024 100 60 z 64 20
  run $n cshim d 0; run $n cshim z 16; run $n p2p
done
run 4 refcounts
run 3 solve 700 60 40 d 0 20; run 3 cshim z 0; run 3 reference_units z 0; run 3 p2p; run 3 pseudo_solve_real 0
