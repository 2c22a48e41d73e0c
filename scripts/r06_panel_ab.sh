cd $GRAFT_REPO_ROOT
CHASE_HIP_TILE_GROUP=2 python3 scripts/dev_gemm_only.py z 65536 2560 3 2>&1 | tail -1 | cut -c88-130 | sed "s/^/full GR=2: /"
python3 scripts/dev_gemm_only.py z 65536 2560 3 2>&1 | tail -1 | cut -c88-130 | sed "s/^/full rule(4): /"
CHASE_HIP_GEMM3M=0 python3 scripts/dev_gemm_only.py z 65536 2560 2 2>&1 | tail -1 | cut -c88-130 | sed "s/^/full 4M: /"
for g in 2 4 2 4; do CHASE_HIP_TILE_GROUP=$g python3 scripts/dev_panel_only.py z N 16384 32768 256 4 30 2>&1 | tail -1 | cut -c1-100 | sed "s/^/GR=$g /"; done
for g in 2 4; do CHASE_HIP_TILE_GROUP=$g python3 scripts/dev_panel_only.py z C 16384 32768 256 4 30 2>&1 | tail -1 | cut -c1-100 | sed "s/^/GR=$g /"; done
