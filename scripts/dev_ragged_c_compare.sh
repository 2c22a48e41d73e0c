REPO=${GRAFT_REPO_ROOT:-$(pwd)}
# 'base' must be built with -DCHASE_NO_RAGGED_C_UNROLL=1 as variant 'noUnroll' to repeat the round-4 comparison; the product is the unrolled form
for round in 1 2; do for v in base noUnroll; do
  if [ $v = base ]; then unset CHASE_HIP_LIB; else export CHASE_HIP_LIB=$REPO/chase_amd/lib/variants/libchase_hip_$v.so; fi
  for n in 133 300 2500; do echo -n "round $round $v n=$n: "; python3 $REPO/scripts/dev_gemm_only.py z 32768 $n 5 C 2>/dev/null | tail -1; done
done; done
