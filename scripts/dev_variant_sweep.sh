#!/bin/bash
# usage: scripts/dev_variant_sweep.sh <variant names...>   ("base" = the product library); cfg2-size filter HEMM, 20 launches
for v in "$@"; do
  if [ $v = base ]; then unset CHASE_HIP_LIB; else export CHASE_HIP_LIB=$PWD/chase_amd/lib/variants/libchase_hip_$v.so; fi
  echo "== $v: $(python3 scripts/dev_gemm_only.py z 16384 640 20 2>&1 | tail -1)"
done
