#!/bin/bash
# What would the full-width filter launch gain if an operand's re-reads never left the L2?  Diagnostic builds in which every
# workgroup streams the SAME A row panel (sameA), the same B column panel (sameB) or both (results wrong on purpose, timing
# only), interleaved with the product build on ONE device.  Build first: scripts/dev_build_variant.sh sameA -DCHASE_DIAG_SAME_A=1 ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for v in base sameA sameB sameAB; do
    if [ $v = base ]; then unset CHASE_HIP_LIB; else export CHASE_HIP_LIB=$REPO/chase_amd/lib/variants/libchase_hip_$v.so; fi
    echo -n "round $round $v: "; python3 $REPO/scripts/dev_gemm_only.py z 65536 2560 3 | tail -1
  done
done
