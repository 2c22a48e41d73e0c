// factor_kernels.hip — LDS-resident diagonal-block Cholesky + triangular inverse (the serial core of the blocked
// device POTRF / TRSM used by CholeskyQR).
//
// Reference call sites replaced: lapackpp::t_potrf('U') and blaspp::t_trsm('R','U','N','N') in
// linalg/internal/cpu/cholqr1.hpp:41-189 (cusolverDnTpotrf / cublasTtrsm in linalg/internal/cuda/cholqr.hpp:110-132).
//
// One 256-thread workgroup factors a nb x nb (nb <= 64) Hermitian diagonal block held entirely in LDS:
//   A_jj = R^H R (upper), R written back in place, and T = R^{-1} (upper) written to a 64-ld workspace block.
// The blocked driver (capi_blas.hip) then forms the row panel and the trailing update / the TRSM sweeps with the
// MFMA GEMM, so every O(n^3) / O(m n^2) flop runs on the matrix cores and only this O(nb^3) core is scalar.
// LAPACK info semantics: the first non-positive (or NaN) pivot at global index p sets *info = p + 1 (first failure wins).

#include <hip/hip_runtime.h>
#include "kernels.h"

namespace chase_hip {

constexpr int FNB = 64;

struct cd { double x, y; };
__device__ __forceinline__ cd cmul_conj_a(cd a, cd b) { return cd{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x}; } // conj(a)*b
__device__ __forceinline__ cd cmul(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

template <bool CPLX>
__global__ __launch_bounds__(256) void potf2_trtri_kernel(double* __restrict__ A, long lda, int nb, int joff,
                                                          double* __restrict__ Tinv, int* __restrict__ info)
{
    using E = typename std::conditional<CPLX, cd, double>::type;
    __shared__ E s[FNB * FNB];          // column-major, ld = 64 (64 KiB complex / 32 KiB real)
    __shared__ double dinv[FNB];
    __shared__ int failed;
    const int tid = threadIdx.x;
    E* Ag = (E*)A;

    if (tid == 0) failed = 0;
    for (int e = tid; e < FNB * FNB; e += 256) {
        const int i = e & 63, j = e >> 6;
        E v;
        if constexpr (CPLX) v = cd{0.0, 0.0}; else v = 0.0;
        if (i < nb && j < nb && i <= j) v = Ag[(long)j * lda + i];
        s[e] = v;
    }
    __syncthreads();

    // ---- unblocked right-looking Cholesky of the upper triangle ----
    for (int k = 0; k < nb; ++k) {
        double akk;
        if constexpr (CPLX) akk = s[k + k * FNB].x; else akk = s[k + k * FNB];
        if (!(akk > 0.0)) {                       // also catches NaN
            if (tid == 0) { failed = 1; atomicCAS(info, 0, joff + k + 1); }
            break;                                // uniform: every thread read the same akk
        }
        const double r = sqrt(akk), rinv = 1.0 / r;
        __syncthreads();                          // everyone has read akk before it is overwritten
        if (tid == 0) {
            if constexpr (CPLX) s[k + k * FNB] = cd{r, 0.0}; else s[k + k * FNB] = r;
            dinv[k] = rinv;
        }
        for (int j = k + 1 + tid; j < nb; j += 256) {
            if constexpr (CPLX) { cd v = s[k + j * FNB]; s[k + j * FNB] = cd{v.x * rinv, v.y * rinv}; }
            else s[k + j * FNB] *= rinv;
        }
        __syncthreads();
        const int rem = nb - k - 1;
        for (int e = tid; e < rem * rem; e += 256) {
            const int i = k + 1 + e % rem, j = k + 1 + e / rem;
            if (i <= j) {
                if constexpr (CPLX) {
                    const cd p = cmul_conj_a(s[k + i * FNB], s[k + j * FNB]);
                    cd v = s[i + j * FNB];
                    s[i + j * FNB] = cd{v.x - p.x, v.y - p.y};
                } else {
                    s[i + j * FNB] -= s[k + i * FNB] * s[k + j * FNB];
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    const bool bad = (failed != 0);

    // ---- write R back (upper triangle only, like LAPACK) ----
    if (!bad) {
        for (int e = tid; e < FNB * FNB; e += 256) {
            const int i = e & 63, j = e >> 6;
            if (i < nb && j < nb && i <= j) Ag[(long)j * lda + i] = s[e];
        }
    }

    // ---- T = R^{-1}: strictly-upper X stored transposed in the strictly-lower part of the tile ----
    if (!bad && tid < nb) {
        const int j = tid;
        const double xjj = dinv[j];
        for (int i = j - 1; i >= 0; --i) {
            if constexpr (CPLX) {
                cd sum = s[i + j * FNB]; sum = cd{sum.x * xjj, sum.y * xjj};
                for (int l = i + 1; l < j; ++l) { const cd p = cmul(s[i + l * FNB], s[j + l * FNB]); sum.x += p.x; sum.y += p.y; }
                s[j + i * FNB] = cd{-sum.x * dinv[i], -sum.y * dinv[i]};
            } else {
                double sum = s[i + j * FNB] * xjj;
                for (int l = i + 1; l < j; ++l) sum += s[i + l * FNB] * s[j + l * FNB];
                s[j + i * FNB] = -sum * dinv[i];
            }
        }
    }
    __syncthreads();
    E* Tg = (E*)Tinv;
    for (int e = tid; e < FNB * FNB; e += 256) {
        const int i = e & 63, j = e >> 6;
        E v;
        if constexpr (CPLX) v = cd{0.0, 0.0}; else v = 0.0;
        if (!bad && i < nb && j < nb) {
            if (i < j) v = s[j + i * FNB];
            else if (i == j) { if constexpr (CPLX) v = cd{dinv[i], 0.0}; else v = dinv[i]; }
        }
        Tg[e] = v;
    }
}

// Batched inverse of the 64x64 upper-triangular diagonal blocks of R (n x n): Tinv[b] = R_bb^{-1}, one workgroup each.
template <bool CPLX>
__global__ __launch_bounds__(256) void trtri_diag_kernel(const double* __restrict__ R, long ldr, int n,
                                                         double* __restrict__ Tinv)
{
    using E = typename std::conditional<CPLX, cd, double>::type;
    __shared__ E s[FNB * FNB];
    __shared__ double dinv[FNB];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int j0 = b * FNB;
    const int nb = min(FNB, n - j0);
    const E* Rg = (const E*)R + (long)j0 * ldr + j0;
    for (int e = tid; e < FNB * FNB; e += 256) {
        const int i = e & 63, j = e >> 6;
        E v;
        if constexpr (CPLX) v = cd{0.0, 0.0}; else v = 0.0;
        if (i < nb && j < nb && i <= j) v = Rg[(long)j * ldr + i];
        s[e] = v;
    }
    __syncthreads();
    if (tid < nb) {
        // 1 / r_jj with a complex-safe reciprocal (the diagonal of a Cholesky factor is real, but stay general)
        if constexpr (CPLX) dinv[tid] = 1.0 / s[tid + tid * FNB].x; else dinv[tid] = 1.0 / s[tid + tid * FNB];
    }
    __syncthreads();
    if (tid < nb) {
        const int j = tid;
        const double xjj = dinv[j];
        for (int i = j - 1; i >= 0; --i) {
            if constexpr (CPLX) {
                cd sum = s[i + j * FNB]; sum = cd{sum.x * xjj, sum.y * xjj};
                for (int l = i + 1; l < j; ++l) { const cd p = cmul(s[i + l * FNB], s[j + l * FNB]); sum.x += p.x; sum.y += p.y; }
                s[j + i * FNB] = cd{-sum.x * dinv[i], -sum.y * dinv[i]};
            } else {
                double sum = s[i + j * FNB] * xjj;
                for (int l = i + 1; l < j; ++l) sum += s[i + l * FNB] * s[j + l * FNB];
                s[j + i * FNB] = -sum * dinv[i];
            }
        }
    }
    __syncthreads();
    E* Tg = (E*)Tinv + (long)b * FNB * FNB;
    for (int e = tid; e < FNB * FNB; e += 256) {
        const int i = e & 63, j = e >> 6;
        E v;
        if constexpr (CPLX) v = cd{0.0, 0.0}; else v = 0.0;
        if (i < nb && j < nb) {
            if (i < j) v = s[j + i * FNB];
            else if (i == j) { if constexpr (CPLX) v = cd{dinv[i], 0.0}; else v = dinv[i]; }
        }
        Tg[e] = v;
    }
}

int trtri_diag(hipStream_t st, bool cplx, const double* R, long ldr, int n, double* Tinv)
{
    const int nblk = (n + FNB - 1) / FNB;
    if (nblk <= 0) return 0;
    if (cplx) hipLaunchKernelGGL(trtri_diag_kernel<true>, dim3(nblk), dim3(256), 0, st, R, ldr, n, Tinv);
    else      hipLaunchKernelGGL(trtri_diag_kernel<false>, dim3(nblk), dim3(256), 0, st, R, ldr, n, Tinv);
    return (int)hipGetLastError();
}

int potf2_trtri(hipStream_t st, bool cplx, double* A, long lda, int nb, int joff, double* Tinv, int* info_dev)
{
    if (cplx) hipLaunchKernelGGL(potf2_trtri_kernel<true>, dim3(1), dim3(256), 0, st, A, lda, nb, joff, Tinv, info_dev);
    else      hipLaunchKernelGGL(potf2_trtri_kernel<false>, dim3(1), dim3(256), 0, st, A, lda, nb, joff, Tinv, info_dev);
    return (int)hipGetLastError();
}

} // namespace chase_hip
