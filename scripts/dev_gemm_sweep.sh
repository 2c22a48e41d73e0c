for n in 640 512 400 320 225 200 133 100 70 40; do python scripts/dev_gemm_only.py z 16384 $n 5 N 2>&1 | tail -1; done
for n in 1280 700 300 133; do python scripts/dev_gemm_only.py d 32768 $n 5 N 2>&1 | tail -1; done
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -3
