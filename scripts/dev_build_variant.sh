#!/bin/bash
# Kernel-development helper: builds chase_amd/lib/variants/libchase_hip_<name>.so = the product library with
# gemm_mfma_f64.hip recompiled under extra -D flags (run `make` first).  Use it with CHASE_HIP_LIB=<that file>.
# usage: scripts/dev_build_variant.sh <name> [-DFOO=1 ...]
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/variants chase_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I/opt/rocm/include -Ichase_amd/csrc -Ichase_amd/host \
    -Wno-unused-result -Wno-unused-value "$@" -c chase_amd/csrc/gemm_mfma_f64.hip -o build/variants/gemm_$NAME.o
OBJS=$(find build/chase_amd -name '*.o' ! -name 'gemm_mfma_f64.hip.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o chase_amd/lib/variants/libchase_hip_$NAME.so build/variants/gemm_$NAME.o $OBJS \
    -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -ldl -lpthread
echo chase_amd/lib/variants/libchase_hip_$NAME.so
