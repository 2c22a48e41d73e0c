# width sweep of the filter HEMM at the cfg2 operator size (complex N = 16384, real N = 32768): uniform ragged tiling on / off
for u in 0 1; do
  for n in 40 100 133 200 300 400 520 640; do CHASE_HIP_UNIFORM_TILES=$u python scripts/dev_gemm_only.py z 16384 $n 5 | sed "s/^/uniform=$u /"; done
  for n in 40 133 200 300 640 1280; do CHASE_HIP_UNIFORM_TILES=$u python scripts/dev_gemm_only.py d 32768 $n 5 | sed "s/^/uniform=$u /"; done
done
