# width sweep of the real filter HEMM (N = 32768): 128-wide tiles only (CHASE_HIP_REAL_NARROW=0) vs narrow 128 x 64 tiles for
# blocks <= 64 columns and the ragged rest (1, default) vs forced rest on narrow tiles (2)
for mode in 0 1 2; do
  for n in 16 40 48 64 100 133 200 300 640 1280; do
    CHASE_HIP_REAL_NARROW=$mode python scripts/dev_gemm_only.py d 32768 $n 5 | sed "s/^/narrow=$mode /"
  done
  for n in 40 133; do
    CHASE_HIP_REAL_NARROW=$mode python scripts/dev_gemm_only.py d 32768 $n 5 C | sed "s/^/narrow=$mode /"
  done
done
