#!/bin/bash
# Experiment: real RCCL collectives between rank PROCESSES that share the box's one GPU, by giving every rank its own NCCL_HOSTID
# (tests/dist_worker.py, CHASE_TEST_FAKE_HOSTS=1): RCCL then treats the ranks as different hosts and uses its socket transport.
# usage: scripts/dev_rccl_fake_hosts.sh <nranks> <scenario> [args...]     (log: gpurun_out/rccl_fake_hosts_<scenario>.log)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
SCEN=$1
export CHASE_TEST_FAKE_HOSTS=1 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4 NCCL_DEBUG=${NCCL_DEBUG:-WARN}
mkdir -p $REPO/gpurun_out
timeout -k 10 ${FAKE_TIMEOUT:-240} python -m torch.distributed.run --nnodes=1 --nproc-per-node=$N --master-addr 127.0.0.1 --master-port ${PORT:-29655} \
    $REPO/tests/dist_worker.py rccl "$@" > $REPO/gpurun_out/rccl_fake_hosts_$SCEN.log 2>&1
rc=$?
echo "fake-hosts $N ranks $* -> rc=$rc"
grep -E "DIST_WORKER_OK|NCCL WARN|Duplicate|Error|error|via NET|Channel 00|NOTE|note:|pseudo solve:" $REPO/gpurun_out/rccl_fake_hosts_$SCEN.log | head -${LINES_SHOWN:-12}
exit $rc
