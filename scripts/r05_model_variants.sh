#!/bin/bash
# Round 5: what this round's changes to the distributed path are worth against MODELLED collectives (4x2 replay of config 4,
# 50 GB/s bus bandwidth, 20 us latency, 32 RCCL-sized workgroups per collective): defaults vs round 4's behaviours.
# (profiles/r05_replay_4x2_model_variants.json was taken when two communication streams were the default: its "defaults" row has
# comm_streams = 2 and its "one_stream" row is today's default.)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r05v}
TAPE=$REPO/profiles/r05_cfg4_tape.npz
cd $REPO
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py --replay-rank 4x2 --tape $TAPE --loopback-busbw ${BW:-50} "$@" > ${OUT}_$name.json 2> ${OUT}_$name.log || { tail -5 ${OUT}_$name.log; exit 1; }
  grep "T_rank" ${OUT}_$name.log | cut -c1-170
}
run defaults X=1 --
run r4_agree_vectors_one_stream CHASE_HIP_RR_AGREE=vectors CHASE_HIP_COMM_STREAMS=1 --
run two_streams X=1 -- --replay-comm-streams 2
run no_pipeline X=1 -- --replay-no-pipeline
run panel128 X=1 -- --replay-panel 128
run panel512 X=1 -- --replay-panel 512
