// kernels.h — internal C++ declarations of the HIP kernel launchers (not part of the C ABI; see include/chase_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace chase_hip {

// C = alpha*op(A)*B + beta*C, column-major, device pointers. cplx: elements are interleaved (re,im) doubles.
// lda/ldb/ldc in elements. alpha/beta point to 1 (real) or 2 (complex) host doubles.
// ws/ws_bytes: optional device workspace for deterministic split-K.
int gemm_f64(hipStream_t st, bool cplx, char opA, int m, int n, int k, const double* alpha, const double* A, long lda,
             const double* B, long ldb, const double* beta, double* C, long ldc, double* ws, size_t ws_bytes,
             int num_cu);

int mfma_f64_peak(hipStream_t st, double* out, int blocks, int iters);
int stream_copy(hipStream_t st, void* dst, const void* src, size_t bytes);

} // namespace chase_hip
