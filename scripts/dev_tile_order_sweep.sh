for g in 1 2 4 8; do for pad in 0 24; do CHASE_HIP_TILE_GROUP=$g python scripts/dev_gemm_only.py z 65536 2560 3 N $pad; done; done
for g in 1 4; do for pad in 0 24; do CHASE_HIP_TILE_GROUP=$g python scripts/dev_gemm_only.py z 16384 640 5 N $pad; done; done
