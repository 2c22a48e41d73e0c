# width sweep of the filter HEMM, round 3: real N = 32768 with the 128 x 64 narrow tile off / on (CHASE_HIP_REAL_NARROW=0 is the
# round-2 scheme: 128-wide tiles + a 128-wide ragged launch for the rest), complex N = 16384 (3M kernel) for reference
for mode in 0 1; do
  for n in 16 40 48 64 100 133 160 200 300 428 640 1200 1280; do
    CHASE_HIP_REAL_NARROW=$mode python scripts/dev_gemm_only.py d 32768 $n 5 | sed "s/^/narrow=$mode /"
  done
  for n in 40 133 300; do
    CHASE_HIP_REAL_NARROW=$mode python scripts/dev_gemm_only.py d 32768 $n 5 C | sed "s/^/narrow=$mode /"
  done
done
for n in 40 100 133 200 300 400 520 640; do python scripts/dev_gemm_only.py z 16384 $n 5; done
