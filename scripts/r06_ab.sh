#!/bin/bash
# A/B of the filter kernel's variants on whatever device the lease has (devices of the pool run the 3M load at different clocks):
# plane-fed loop (default), the same with tile group 4, prefetch depth 1 (variant library), round 2-5's loop (variant library), 4M.
# Build the variant libraries first: scripts/dev_build_variant.sh m3old -DCHASE_M3_SPLANE=0; scripts/dev_build_variant.sh m3depth1 -DCHASE_M3_DEPTH=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
V=chase_amd/lib/variants
run() { python3 scripts/dev_gemm_only.py z 65536 2560 3 2>&1 | tail -1 | sed "s/^/$1: /"; }
for i in 1 2; do
run "new (rule)   "
CHASE_HIP_TILE_GROUP=2 run "new GR=2     "
CHASE_HIP_TILE_GROUP=4 run "new GR=4     "
CHASE_HIP_LIB=$V/libchase_hip_m3depth1.so run "depth 1      "
CHASE_HIP_LIB=$V/libchase_hip_m3depth1.so CHASE_HIP_TILE_GROUP=4 run "depth 1 GR=4 "
CHASE_HIP_LIB=$V/libchase_hip_m3old.so run "old          "
done
CHASE_HIP_GEMM3M=0 run "4M           "
