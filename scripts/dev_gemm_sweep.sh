#!/bin/bash
# width sweep of the filter-shaped HEMM (timing by HIP events inside scripts/dev_gemm_only.py): complex 3M (default), complex 4M, real
echo "## complex N=16384, op N, 3M (filter default)"
for n in 2560 1280 640 512 400 320 225 200 133 100 70 40 16; do python3 scripts/dev_gemm_only.py z 16384 $n 5 N 2>&1 | tail -1; done
echo "## complex N=16384, op C, 3M"
for n in 640 133; do python3 scripts/dev_gemm_only.py z 16384 $n 5 C 2>&1 | tail -1; done
echo "## complex N=16384, op N, 4M (CHASE_HIP_GEMM3M=0)"
for n in 2560 640 400 225 133 40; do CHASE_HIP_GEMM3M=0 python3 scripts/dev_gemm_only.py z 16384 $n 5 N 2>&1 | tail -1; done
echo "## real N=32768, op N / op C"
for n in 2560 1280 700 300 133 40; do python3 scripts/dev_gemm_only.py d 32768 $n 5 N 2>&1 | tail -1; done
python3 scripts/dev_gemm_only.py d 32768 1280 5 C 2>&1 | tail -1
echo "## MFMA issue-rate probe"
python3 -c "
import sys; sys.path.insert(0,'.')
from chase_amd.capi import Context
c=Context(0); print('v_mfma_f64_16x16x4_f64 register-resident probe: %.2f TFLOP/s' % c.mfma_f64_peak())"
