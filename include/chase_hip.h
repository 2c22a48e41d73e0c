/* chase_hip.h — C ABI of the MI355X-native ChASE hot-path backend (libchase_hip.so)
 *
 * Boundary: the reference solver talks to a backend ONLY through the virtuals of
 * chase::ChaseBase<T> (/root/reference algorithm/interface.hpp:46-434).  This header is the thin C ABI the
 * C++ Impl class (chase_amd/host/chase_hip_impl.hpp) calls, and what a ChASE maintainer would bind from an
 * Impl/chase_hip plugin (INTEGRATION.md shows the stub).  Plain pointers and sizes only, no C++/torch types.
 *
 * Conventions
 *   - every entry point returns int: 0 = ok, >0 = LAPACK-style info (potrf), <0 = runtime error
 *     (CHASE_HIP_E* or -(hipError_t)); no exceptions cross the ABI.  chase_hip_last_error() gives text.
 *   - matrices are column-major; complex = interleaved (re,im) doubles ("z" entry points take void* / double[2]).
 *   - "dev" pointers are HIP device pointers; "host" pointers are ordinary host memory.
 *   - all device work is enqueued on the context's HIP stream; entry points that return scalars to the host
 *     synchronise that stream themselves, the others are asynchronous.
 */
#ifndef CHASE_HIP_H
#define CHASE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHASE_HIP_OK 0
#define CHASE_HIP_EINVAL (-1001)   /* bad argument (shape, pointer, op) */
#define CHASE_HIP_ENODEV (-1002)   /* no usable HIP device / wrong architecture */
#define CHASE_HIP_ENOMEM (-1003)
#define CHASE_HIP_ENOTCONV (-1004) /* host eigensolver / iteration failed to converge */
#define CHASE_HIP_ECOMM (-1005)    /* RCCL / transport error */
#define CHASE_HIP_ELAPACK (-1006)  /* host LAPACK provider missing */

typedef struct chase_hip_ctx chase_hip_ctx;

/* ---- context -------------------------------------------------------------------------------------------------- */
/* device: HIP ordinal.  stream: an existing hipStream_t to enqueue on (e.g. torch's current stream) or NULL to let the
 * context create its own non-blocking stream. */
int chase_hip_ctx_create(chase_hip_ctx** out, int device, void* stream);
int chase_hip_ctx_destroy(chase_hip_ctx* ctx);
int chase_hip_ctx_sync(chase_hip_ctx* ctx);
void* chase_hip_ctx_stream(chase_hip_ctx* ctx);
const char* chase_hip_last_error(void);
/* name may be NULL.  clock_khz is the reported max engine clock. */
int chase_hip_device_info(chase_hip_ctx* ctx, int* num_cu, int* clock_khz, size_t* hbm_bytes, char* name,
                          int name_len);
const char* chase_hip_version(void);

/* ---- device memory plumbing ----------------------------------------------------------------------------------- */
int chase_hip_malloc(chase_hip_ctx* ctx, void** dev, size_t bytes);
int chase_hip_free(chase_hip_ctx* ctx, void* dev);
int chase_hip_memcpy_h2d(chase_hip_ctx* ctx, void* dev, const void* host, size_t bytes); /* synchronous */
int chase_hip_memcpy_d2h(chase_hip_ctx* ctx, void* host, const void* dev, size_t bytes); /* synchronous */
int chase_hip_memcpy_d2d(chase_hip_ctx* ctx, void* dst, const void* src, size_t bytes);  /* stream-ordered */
int chase_hip_memset(chase_hip_ctx* ctx, void* dev, int value, size_t bytes);            /* stream-ordered */
/* stream-ordered timing helpers: elapsed milliseconds of the work enqueued by fn-free bracketing */
int chase_hip_timer_start(chase_hip_ctx* ctx);
int chase_hip_timer_stop(chase_hip_ctx* ctx, float* ms); /* synchronises */

/* ---- level-3 kernels (device pointers) -------------------------------------------------------------------------
 * C = alpha*op(A)*B + beta*C, opA in {'N','C'} ('T' == 'C' for real).  Replaces the reference's
 * blaspp::t_gemm / cublasTgemm call sites: Impl/chase_cpu/chase_cpu.hpp:497-504, Impl/chase_gpu/chase_gpu.hpp:668-675,
 * linalg/internal/mpi/hemm.hpp:155-167,209-221, linalg/internal/cpu/rayleighRitz.hpp:84-88,110. */
int chase_hip_gemm_d(chase_hip_ctx* ctx, char opA, int m, int n, int k, double alpha, const double* A, long lda,
                     const double* B, long ldb, double beta, double* C, long ldc);
int chase_hip_gemm_z(chase_hip_ctx* ctx, char opA, int m, int n, int k, const double alpha[2], const void* A,
                     long lda, const void* B, long ldb, const double beta[2], void* C, long ldc);

/* register-resident v_mfma_f64_16x16x4_f64 issue-rate probe: returns achieved TFLOP/s */
int chase_hip_mfma_f64_peak(chase_hip_ctx* ctx, double* tflops);
/* streaming-copy probe: achieved HBM GB/s for a bytes-sized device-to-device float4 copy */
int chase_hip_hbm_copy_peak(chase_hip_ctx* ctx, size_t bytes, double* gbps);

#ifdef __cplusplus
}
#endif
#endif /* CHASE_HIP_H */
