// hetrd.hip — Hermitian eigensolver for the projected matrix with the O(n^3) work on the GPU:
//     A = Q T Q^H  (Householder tridiagonalisation, device)  ->  T = Z L Z^T (LAPACK stemr on the host, O(n^2))
//     ->  eigenvectors Q Z (blocked compact-WY back-transformation with the MFMA GEMM, device).
//
// Replaces the host HEEVD of the Rayleigh-Ritz step (reference: lapackpp::t_heevd, linalg/internal/cpu/rayleighRitz.hpp:104;
// cusolverDnTheevd, linalg/internal/nccl/rayleighRitz.hpp:170-173) for n >= 256, where a replicated host HEEVD is the
// strong-scaling bottleneck (n = 2560: 1.3 s on 16 host threads per iteration vs 0.18 s of filter GEMM per step).
// Algorithm = LAPACK xHETD2 ('L') + xUNMTR; per column four multi-workgroup launches, all scalars stay on the device:
//   larfg      reflector of column k (single workgroup, wave-shuffle norm)
//   gemv       partial products of the trailing block with v over 32-column chunks   (HBM-bound, one pass over A22)
//   reduce     p = tau * sum(partials), per-workgroup partial dot p^H v
//   her2       A22 -= v w^H + w v^H with w = p - (tau/2)(p^H v) v formed on the fly (HBM-bound, one read+write of A22)
// The full (both triangles) trailing block is kept up to date so that the products are plain coalesced GEMVs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/chase_hip.h"
#include "ctx.h"
#include "kernels.h"
#include "host_lapack.h"

namespace chase_hip {

constexpr int TRB = 256;   // rows per workgroup
constexpr int TCW = 32;    // columns per gemv chunk
constexpr int UCW = 16;    // columns per her2 chunk

__device__ __forceinline__ double tsum(double v, double* sm)
{
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// column k: x = A[k+1:n, k].  v (v[0] = 1) -> vbuf[0:m], A[k+2:, k] keeps v[1:], e[k] = beta, d[k] = Re A[k,k], tau[k]
template <bool CPLX>
__global__ __launch_bounds__(256) void trd_larfg_kernel(double* __restrict__ A, long lda, int n, int k,
                                                        double* __restrict__ vbuf, double* __restrict__ d,
                                                        double* __restrict__ e, double* __restrict__ tau)
{
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double sm[4];
    const int m = n - k - 1;
    double* x = A + ((long)k * lda + k + 1) * E;
    double s = 0.0;
    for (int i = E + threadIdx.x; i < m * E; i += 256) { const double t = x[i]; s += t * t; }
    const double xn2 = tsum(s, sm);
    const double ar = x[0], ai = CPLX ? x[1] : 0.0;
    double beta, tr, ti, sr, si;
    if (xn2 == 0.0 && ai == 0.0) { beta = ar; tr = ti = 0.0; sr = si = 0.0; }
    else {
        const double nrm = sqrt(ar * ar + ai * ai + xn2);
        beta = (ar >= 0.0) ? -nrm : nrm;
        tr = (beta - ar) / beta; ti = -ai / beta;
        const double dr = ar - beta, di = ai, den = dr * dr + di * di;
        sr = dr / den; si = -di / den;
    }
    __syncthreads();
    for (int i = 1 + threadIdx.x; i < m; i += 256) {
        double vr, vi = 0.0;
        if constexpr (CPLX) {
            const double xr = x[2 * i], xi = x[2 * i + 1];
            vr = xr * sr - xi * si; vi = xr * si + xi * sr;
            x[2 * i] = vr; x[2 * i + 1] = vi;
            vbuf[2 * i] = vr; vbuf[2 * i + 1] = vi;
        } else {
            vr = x[i] * sr; x[i] = vr; vbuf[i] = vr;
        }
    }
    if (threadIdx.x == 0) {
        vbuf[0] = 1.0; if (CPLX) vbuf[1] = 0.0;
        x[0] = beta; if (CPLX) x[1] = 0.0;                 // sub-diagonal entry of T (real)
        e[k] = beta;
        d[k] = A[((long)k * lda + k) * E];
        tau[k * E] = tr; if (CPLX) tau[k * E + 1] = ti;
        if (k == n - 2) d[n - 1] = A[((long)(n - 1) * lda + (n - 1)) * E];   // last diagonal is not touched: m == 1
    }
}

// part[cc][i] = sum_{j in chunk cc} A22[i, j] * v[j],  A22 = A[k+1:, k+1:] (m x m)
template <bool CPLX>
__global__ __launch_bounds__(256) void trd_gemv_kernel(const double* __restrict__ A, long lda, int n, int k,
                                                       const double* __restrict__ vbuf, double* __restrict__ part)
{
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double vs[TCW * 2];
    const int m = n - k - 1;
    const int i = blockIdx.x * TRB + threadIdx.x;
    const int j0 = blockIdx.y * TCW;
    const int jn = min(TCW, m - j0);
    if (threadIdx.x < jn * E) vs[threadIdx.x] = vbuf[(long)j0 * E + threadIdx.x];
    __syncthreads();
    if (i >= m) return;
    const double* a = A + ((long)(k + 1 + j0) * lda + (k + 1) + i) * E;
    double pr = 0.0, pi = 0.0;
    for (int j = 0; j < jn; ++j) {
        if constexpr (CPLX) {
            const double xr = a[0], xi = a[1], vr = vs[2 * j], vi = vs[2 * j + 1];
            pr += xr * vr - xi * vi; pi += xr * vi + xi * vr;
        } else {
            pr += a[0] * vs[j];
        }
        a += lda * E;
    }
    double* o = part + ((long)blockIdx.y * m + i) * E;
    o[0] = pr; if (CPLX) o[1] = pi;
}

// p = tau * sum_cc part[cc];  dots[blockIdx.x] = sum over this workgroup's rows of conj(p_i) v_i
template <bool CPLX>
__global__ __launch_bounds__(256) void trd_reduce_kernel(const double* __restrict__ part, int nchunks, int n, int k,
                                                         const double* __restrict__ vbuf, const double* __restrict__ tau,
                                                         double* __restrict__ pbuf, double* __restrict__ dots)
{
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double sm[4];
    const int m = n - k - 1;
    const int i = blockIdx.x * TRB + threadIdx.x;
    const double tr = tau[k * E], ti = CPLX ? tau[k * E + 1] : 0.0;
    double dr = 0.0, di = 0.0;
    if (i < m) {
        double sr = 0.0, si = 0.0;
        for (int c = 0; c < nchunks; ++c) {
            const double* q = part + ((long)c * m + i) * E;
            sr += q[0]; if (CPLX) si += q[1];
        }
        const double pr = tr * sr - ti * si, pi = tr * si + ti * sr;
        pbuf[(long)i * E] = pr; if (CPLX) pbuf[(long)i * E + 1] = pi;
        const double vr = vbuf[(long)i * E], vi = CPLX ? vbuf[(long)i * E + 1] : 0.0;
        dr = pr * vr + pi * vi;          // conj(p) * v
        di = pr * vi - pi * vr;
    }
    dr = tsum(dr, sm);
    if (CPLX) di = tsum(di, sm);
    if (threadIdx.x == 0) { dots[blockIdx.x * 2] = dr; dots[blockIdx.x * 2 + 1] = di; }
}

// A22[i,j] -= v_i conj(w_j) + w_i conj(v_j),  w = p + alpha v,  alpha = -(tau/2) * (p^H v)
template <bool CPLX>
__global__ __launch_bounds__(256) void trd_her2_kernel(double* __restrict__ A, long lda, int n, int k,
                                                       const double* __restrict__ vbuf, const double* __restrict__ pbuf,
                                                       const double* __restrict__ tau, const double* __restrict__ dots,
                                                       int ndots)
{
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double vs[UCW * 2], ws[UCW * 2];
    const int m = n - k - 1;
    // alpha (every workgroup re-sums the few partial dots in a fixed order)
    double qr = 0.0, qi = 0.0;
    for (int b = 0; b < ndots; ++b) { qr += dots[2 * b]; qi += dots[2 * b + 1]; }
    const double tr = tau[k * E], ti = CPLX ? tau[k * E + 1] : 0.0;
    const double alr = -0.5 * (tr * qr - ti * qi), ali = -0.5 * (tr * qi + ti * qr);
    const int j0 = blockIdx.y * UCW;
    const int jn = min(UCW, m - j0);
    if (threadIdx.x < jn) {
        const int j = j0 + threadIdx.x;
        const double vr = vbuf[(long)j * E], vi = CPLX ? vbuf[(long)j * E + 1] : 0.0;
        const double pr = pbuf[(long)j * E], pi = CPLX ? pbuf[(long)j * E + 1] : 0.0;
        vs[2 * threadIdx.x] = vr; vs[2 * threadIdx.x + 1] = vi;
        ws[2 * threadIdx.x] = pr + alr * vr - ali * vi;
        ws[2 * threadIdx.x + 1] = pi + alr * vi + ali * vr;
    }
    __syncthreads();
    const int i = blockIdx.x * TRB + threadIdx.x;
    if (i >= m) return;
    const double vir = vbuf[(long)i * E], vii = CPLX ? vbuf[(long)i * E + 1] : 0.0;
    const double pir = pbuf[(long)i * E], pii = CPLX ? pbuf[(long)i * E + 1] : 0.0;
    const double wir = pir + alr * vir - ali * vii, wii = pii + alr * vii + ali * vir;
    double* a = A + ((long)(k + 1 + j0) * lda + (k + 1) + i) * E;
    for (int j = 0; j < jn; ++j) {
        const double vjr = vs[2 * j], vji = vs[2 * j + 1], wjr = ws[2 * j], wji = ws[2 * j + 1];
        if constexpr (CPLX) {
            // v_i conj(w_j) + w_i conj(v_j)
            a[0] -= (vir * wjr + vii * wji) + (wir * vjr + wii * vji);
            a[1] -= (vii * wjr - vir * wji) + (wii * vjr - wir * vji);
        } else {
            a[0] -= vir * wjr + wir * vjr;
        }
        a += lda * E;
    }
}

// Zc (n x n, T) <- Z (n x n real)
template <bool CPLX>
__global__ __launch_bounds__(256) void real_to_T_kernel(const double* __restrict__ Z, double* __restrict__ Zc, long total)
{
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        if constexpr (CPLX) { Zc[2 * e] = Z[e]; Zc[2 * e + 1] = 0.0; }
        else Zc[e] = Z[e];
    }
}

// declared in hhqr.hip
int hh_apply_q_left(chase_hip_ctx* c, bool cplx, const double* Vstore, long ldv, int m, int nref, const double* tau,
                    double* Cm, long ldc, int ncols);

} // namespace chase_hip

using namespace chase_hip;

#define HK(x)                                                                                                          \
    do {                                                                                                               \
        x;                                                                                                             \
        hipError_t e_ = hipGetLastError();                                                                             \
        if (e_ != hipSuccess) return hip_fail(e_, #x);                                                                 \
    } while (0)
#define HC(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) return hip_fail(e_, #x);                                                                 \
    } while (0)

/* Hermitian eigendecomposition of the device matrix A (n x n, full storage, lower triangle authoritative):
 * eigenvalues ascending to w_host, eigenvectors overwrite A.  Tridiagonalisation and back-transformation on the GPU,
 * the O(n^2) tridiagonal eigenproblem (dstemr) on the host. */
extern "C" int chase_hip_heevd_gpu(chase_hip_ctx* c, int cplx_, int n, void* A_, long lda, double* w_host)
{
    if (!c || !A_ || !w_host) return set_error(CHASE_HIP_EINVAL, "heevd_gpu: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (n < 3 || lda < n) return set_error(CHASE_HIP_EINVAL, "heevd_gpu: need n >= 3 and lda >= n");
    const bool cplx = cplx_ != 0;
    const int E = cplx ? 2 : 1;
    double* A = (double*)A_;
    hipStream_t st = c->stream;
    int rc = 0;
    const int nch_max = (n + TCW - 1) / TCW, nrb_max = (n + TRB - 1) / TRB;
    // scratch: vbuf, pbuf (n T each) | part (nch_max x n T) | dots | d, e (n) | tau (n T) | Z (n x n real) | Zc (n x n T)
    const size_t szv = (size_t)n * E, szpart = (size_t)nch_max * n * E, szZ = (size_t)n * n;
    double* blk = nullptr;
    hipError_t he = hipMalloc((void**)&blk, (2 * szv + szpart + 2 * nrb_max + 2 * (size_t)n + szv + szZ + szZ * E + 64) * sizeof(double));
    if (he != hipSuccess) return set_error(CHASE_HIP_ENOMEM, "heevd_gpu: scratch allocation failed");
    double* vbuf = blk; double* pbuf = vbuf + szv; double* part = pbuf + szv; double* dots = part + szpart;
    double* dd = dots + 2 * nrb_max; double* de = dd + n; double* tau = de + n; double* Zr = tau + szv; double* Zc = Zr + szZ;
    const auto t_start = std::chrono::steady_clock::now();
    auto body = [&]() -> int {
        // symmetrise from the lower triangle so that the full-storage GEMV sees an exactly Hermitian matrix
        {
            int e = mirror_lower(st, A, lda, n, E);
            if (e) return hip_fail((hipError_t)e, "mirror_lower");
        }
        for (int k = 0; k < n - 1; ++k) {
            const int m = n - k - 1;
            if (cplx) HK(hipLaunchKernelGGL(trd_larfg_kernel<true>, dim3(1), dim3(256), 0, st, A, lda, n, k, vbuf, dd, de, tau));
            else      HK(hipLaunchKernelGGL(trd_larfg_kernel<false>, dim3(1), dim3(256), 0, st, A, lda, n, k, vbuf, dd, de, tau));
            if (m < 2) continue;                         // 1 x 1 trailing block: H acts trivially on it (tau handled in Q)
            const int nrb = (m + TRB - 1) / TRB, nch = (m + TCW - 1) / TCW, nuc = (m + UCW - 1) / UCW;
            if (cplx) {
                HK(hipLaunchKernelGGL(trd_gemv_kernel<true>, dim3(nrb, nch), dim3(256), 0, st, A, lda, n, k, vbuf, part));
                HK(hipLaunchKernelGGL(trd_reduce_kernel<true>, dim3(nrb), dim3(256), 0, st, part, nch, n, k, vbuf, tau, pbuf, dots));
                HK(hipLaunchKernelGGL(trd_her2_kernel<true>, dim3(nrb, nuc), dim3(256), 0, st, A, lda, n, k, vbuf, pbuf, tau, dots, nrb));
            } else {
                HK(hipLaunchKernelGGL(trd_gemv_kernel<false>, dim3(nrb, nch), dim3(256), 0, st, A, lda, n, k, vbuf, part));
                HK(hipLaunchKernelGGL(trd_reduce_kernel<false>, dim3(nrb), dim3(256), 0, st, part, nch, n, k, vbuf, tau, pbuf, dots));
                HK(hipLaunchKernelGGL(trd_her2_kernel<false>, dim3(nrb, nuc), dim3(256), 0, st, A, lda, n, k, vbuf, pbuf, tau, dots, nrb));
            }
        }
        // tridiagonal eigenproblem on the host
        std::vector<double> hd(n), hee(n), hz((size_t)n * n);
        HC(hipMemcpyAsync(hd.data(), dd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
        HC(hipMemcpyAsync(hee.data(), de, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToHost, st));
        HC(hipStreamSynchronize(st));
        auto t0b = std::chrono::steady_clock::now();
        hee[n - 1] = 0.0;
        // divide & conquer: threaded GEMMs inside, eigenvectors orthogonal to ~eps (MRRR gives ~1e-13 at n = 2560)
        static const bool use_mrrr = getenv("CHASE_HIP_TRIDIAG_MRRR") != nullptr;
        static const bool dbg = getenv("CHASE_HIP_HEEVD_TIMING") != nullptr;
        auto t1 = std::chrono::steady_clock::now();
        int r2 = use_mrrr ? host_stemr(n, hd.data(), hee.data(), w_host, hz.data(), n)
                          : host_stedc(n, hd.data(), hee.data(), w_host, hz.data(), n);
        auto t2 = std::chrono::steady_clock::now();
        if (r2) return r2;
        HC(hipMemcpyAsync(Zr, hz.data(), szZ * sizeof(double), hipMemcpyHostToDevice, st));
        if (cplx) HK(hipLaunchKernelGGL(real_to_T_kernel<true>, dim3(1024), dim3(256), 0, st, Zr, Zc, (long)szZ));
        else      HK(hipLaunchKernelGGL(real_to_T_kernel<false>, dim3(1024), dim3(256), 0, st, Zr, Zc, (long)szZ));
        // eigenvectors = Q Z with Q = H_0 ... H_{n-2} acting on rows 1..n-1: the reflectors sit in B = A[1:, 0:n-1] in the
        // QR storage convention (unit at B[k,k], tail below)
        r2 = hh_apply_q_left(c, cplx, A + (size_t)1 * E, lda, n - 1, n - 1, tau, Zc + (size_t)1 * E, n, n);
        if (r2) return r2;
        int e2 = copy2d(st, Zc, (long)n * E, A, lda * E, (long)n * E, n);
        if (e2) return hip_fail((hipError_t)e2, "heevd_gpu copy-back");
        HC(hipStreamSynchronize(st));
        if (dbg) {
            auto t3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "heevd_gpu n=%d: tridiag %.1f ms, host tridiagonal eig %.1f ms, back-transform %.1f ms\n", n,
                    ms(t_start, t0b), ms(t1, t2), ms(t2, t3));
        }
        return 0;
    };
    rc = body();
    hipStreamSynchronize(st);
    hipFree(blk);
    return rc;
}
