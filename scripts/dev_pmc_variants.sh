#!/bin/bash
# FETCH_SIZE + time of a full-width config-4 filter HEMM for library variants: scripts/dev_pmc_variants.sh <variant...>
for v in "$@"; do
  if [ $v = base ]; then unset CHASE_HIP_LIB; else export CHASE_HIP_LIB=$PWD/chase_amd/lib/variants/libchase_hip_$v.so; fi
  PMC_GROUPS="FETCH_SIZE" scripts/prof_pmc.sh gpurun_out/pmc_var_$v.txt z 65536 2560 2 > /dev/null 2>&1
  echo "$v (group=${CHASE_HIP_TILE_GROUP:-default}): $(grep FETCH_SIZE gpurun_out/pmc_var_$v.txt | grep -v group | head -1 | awk '{print $3}') | $(grep HEMM /tmp/pmc_0.log | tail -1 | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}')"
done
