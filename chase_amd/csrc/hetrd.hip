// hetrd.hip — Hermitian eigensolver for the projected matrix with the O(n^3) work on the GPU:
//     A = Q T Q^H  (Householder tridiagonalisation, device)  ->  T = Z L Z^T (LAPACK stemr on the host, O(n^2))
//     ->  eigenvectors Q Z (blocked compact-WY back-transformation with the MFMA GEMM, device).
//
// Replaces the host HEEVD of the Rayleigh-Ritz step (reference: lapackpp::t_heevd, linalg/internal/cpu/rayleighRitz.hpp:104;
// cusolverDnTheevd, linalg/internal/nccl/rayleighRitz.hpp:170-173) for n >= 256, where a replicated host HEEVD is the
// strong-scaling bottleneck (n = 2560: 1.3 s on 16 host threads per iteration vs 0.18 s of filter GEMM per step).
// Default: BLOCKED reduction (LAPACK xLATRD / xHETRD 'L' structure, panels of 32 columns, ltrd_* kernels below): one pass
// over the trailing block and three launches per column, the rank-64 trailing update through the MFMA GEMM once per panel.
// n = 2560 complex: tridiagonalisation 104 -> 63 ms (profiles/r02_heevd.txt); what is left is the chain of three dependent
// launches per column (three global reductions: |x|^2, A v, w^H v) at ~8 us each.
// CHASE_HIP_TRD_UNBLOCKED=1 selects the round-1 algorithm = LAPACK xHETD2 ('L'): per column four launches,
//   larfg      reflector of column k (single workgroup, wave-shuffle norm)
//   gemv       partial products of the trailing block with v over 32-column chunks   (HBM-bound, one pass over A22)
//   reduce     p = tau * sum(partials), per-workgroup partial dot p^H v
//   her2       A22 -= v w^H + w v^H with w = p - (tau/2)(p^H v) v formed on the fly (HBM-bound, one read+write of A22)
// Both keep the full (both triangles) trailing block so that the products are plain coalesced GEMVs; all scalars stay on
// the device.
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
    if ((int)threadIdx.x < jn * E) vs[threadIdx.x] = vbuf[(long)j0 * E + threadIdx.x];
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
    if ((int)threadIdx.x < jn) {
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


// =====================================================================================================================
// Blocked tridiagonalisation (LAPACK xLATRD / xHETRD 'L' structure): inside a panel of LNB columns the trailing block is
// NOT updated - the product A22 v is taken with the stale block and corrected with the panel's V and W (m x 2 jj GEMVs) -
// and the rank-2 LNB update A22 -= V W^H + W V^H goes through the MFMA GEMM once per panel.  Per column: ONE pass over
// the trailing block (the unblocked version above makes three) in three light launches - one per global reduction the
// column needs (|x|^2, then A v / V^H v / W^H v, then w^H v):
//   colupd   finalise the previous w, bring column k up to date (a -= V conj(W[k,:])^T + W conj(V[k,:])^T), partial |x|^2
//   gemv     every workgroup derives beta / tau / scale from the partial norms, forms v on the fly, part = A22 v (chunks);
//            extra workgroups of the same launch: V^H v and W^H v (2 jj dot products)
//   w        p = sum(part) - V (W^H v) - W (V^H v), w = tau p, partial w^H v; reflector into A's column
// X = [V | W] is the panel buffer (n rows, global row index, zero above the reflector heads).
#ifndef CHASE_LNB
#define CHASE_LNB 32
#endif
constexpr int LNB = CHASE_LNB;   // panel width of the blocked tridiagonalisation (32; 64 is slower: 71.8 against 63.1 ms at n = 2560, profiles/r04_heevd.txt)
constexpr int LTCW = 128; // largest column chunk of the blocked reduction's GEMV (the host picks 64 or 128: fewer partials to re-add
                          // for large trailing blocks, enough workgroups for small ones)

template <bool CPLX>
__global__ __launch_bounds__(256) void ltrd_colupd_kernel(double* __restrict__ A, long lda, int n, int k, int jj,
                                                          double* __restrict__ X, const double* __restrict__ tau,
                                                          const double* __restrict__ dots, int ndots,
                                                          double* __restrict__ npart, double* __restrict__ d)
{
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double cw[LNB * 2], cv[LNB * 2], sm[4];
    double alr = 0.0, ali = 0.0;                        // w_{jj-1} += alpha v_{jj-1}, alpha = -(tau/2) (w^H v)
    if (jj > 0) {
        double qr = 0.0, qi = 0.0;
        for (int b = 0; b < ndots; ++b) { qr += dots[2 * b]; qi += dots[2 * b + 1]; }
        const double tr = tau[(k - 1) * E], ti = CPLX ? tau[(k - 1) * E + 1] : 0.0;
        alr = -0.5 * (tr * qr - ti * qi); ali = -0.5 * (tr * qi + ti * qr);
    }
    double* V = X;
    double* W = X + (long)LNB * n * E;
    if ((int)threadIdx.x < jj) {
        const int c = threadIdx.x;
        const double vr = V[((long)c * n + k) * E], vi = CPLX ? V[((long)c * n + k) * E + 1] : 0.0;
        double wr = W[((long)c * n + k) * E], wi = CPLX ? W[((long)c * n + k) * E + 1] : 0.0;
        if (c == jj - 1) { wr += alr * vr - ali * vi; wi += alr * vi + ali * vr; }
        cw[2 * c] = wr; cw[2 * c + 1] = -wi;           // conj(W[k, c])
        cv[2 * c] = vr; cv[2 * c + 1] = -vi;           // conj(V[k, c])
    }
    __syncthreads();
    const int r = k + blockIdx.x * TRB + threadIdx.x;
    double s = 0.0;
    if (r < n) {
        double ar = A[((long)k * lda + r) * E], ai = CPLX ? A[((long)k * lda + r) * E + 1] : 0.0;
        for (int c = 0; c < jj; ++c) {
            const double vr = V[((long)c * n + r) * E], vi = CPLX ? V[((long)c * n + r) * E + 1] : 0.0;
            double wr = W[((long)c * n + r) * E], wi = CPLX ? W[((long)c * n + r) * E + 1] : 0.0;
            if (c == jj - 1) {
                wr += alr * vr - ali * vi; wi += alr * vi + ali * vr;
                // write the finalised w back - except in row k: every workgroup reads W[k, jj-1] (raw) for cw above, and
                // nothing after this launch reads that entry (later columns use rows > k), so it must stay raw here
                if (r != k) { W[((long)c * n + r) * E] = wr; if (CPLX) W[((long)c * n + r) * E + 1] = wi; }
            }
            // a -= V[r,c] conj(W[k,c]) + W[r,c] conj(V[k,c])
            ar -= (vr * cw[2 * c] - vi * cw[2 * c + 1]) + (wr * cv[2 * c] - wi * cv[2 * c + 1]);
            ai -= (vr * cw[2 * c + 1] + vi * cw[2 * c]) + (wr * cv[2 * c + 1] + wi * cv[2 * c]);
        }
        if (r == k) { ai = 0.0; d[k] = ar; }            // the diagonal of a Hermitian matrix is real
        A[((long)k * lda + r) * E] = ar; if (CPLX) A[((long)k * lda + r) * E + 1] = ai;
        if (r >= k + 2) s = ar * ar + ai * ai;
    }
    s = tsum(s, sm);
    if (threadIdx.x == 0) npart[blockIdx.x] = s;
}

// scal[0..4] = tau_re, tau_im, beta, scale_re, scale_im of column k (LAPACK xLARFG from the partial norms)
template <bool CPLX>
__device__ __forceinline__ void ltrd_scalars(const double* __restrict__ A, long lda, int k, const double* __restrict__ npart,
                                             int nnp, double& beta, double& tr, double& ti, double& sr, double& si)
{
    constexpr int E = CPLX ? 2 : 1;
    double xn2 = 0.0;
    for (int b = 0; b < nnp; ++b) xn2 += npart[b];
    const double ar = A[((long)k * lda + k + 1) * E], ai = CPLX ? A[((long)k * lda + k + 1) * E + 1] : 0.0;
    if (xn2 == 0.0 && ai == 0.0) { beta = ar; tr = ti = 0.0; sr = si = 0.0; return; }
    const double nrm = sqrt(ar * ar + ai * ai + xn2);
    beta = (ar >= 0.0) ? -nrm : nrm;
    tr = (beta - ar) / beta; ti = -ai / beta;
    const double dr = ar - beta, di = ai, den = dr * dr + di * di;
    sr = dr / den; si = -di / den;
}

// grid (nrb, nch + extra): rows blockIdx.y < nch are the GEMV chunks; the remaining workgroups (linear index b < 2 jj)
// compute coef[b] = X[:, c_b]^H v  (b < jj: c_b = b -> V^H v;  b >= jj: c_b = LNB + b - jj -> W^H v) with the same on-the-fly v
template <bool CPLX>
__global__ __launch_bounds__(256) void ltrd_gemv_kernel(const double* __restrict__ A, long lda, int n, int k, int jj,
                                                        double* __restrict__ X, const double* __restrict__ npart, int nnp,
                                                        double* __restrict__ part, double* __restrict__ scal,
                                                        double* __restrict__ tau, double* __restrict__ e, int nch,
                                                        double* __restrict__ coef, int tcw)
{
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double vs[LTCW * 2];
    __shared__ double sm[4];
    const int m = n - k - 1;
    double beta, tr, ti, sr, si;
    ltrd_scalars<CPLX>(A, lda, k, npart, nnp, beta, tr, ti, sr, si);
    const double* x = A + ((long)k * lda + k + 1) * E;                  // column k below the diagonal (unscaled)
    auto vof = [&](int j, double& vr, double& vi) {
        if (j == 0) { vr = 1.0; vi = 0.0; return; }
        if constexpr (CPLX) { const double xr = x[2 * j], xi = x[2 * j + 1]; vr = xr * sr - xi * si; vi = xr * si + xi * sr; }
        else { vr = x[j] * sr; vi = 0.0; }
    };
    if ((int)blockIdx.y >= nch) {
        const int b = ((int)blockIdx.y - nch) * gridDim.x + blockIdx.x;
        if (b >= 2 * jj) return;
        const int c = (b < jj) ? b : LNB + (b - jj);
        const double* xc = X + ((long)c * n + k + 1) * E;
        double ar_ = 0.0, ai_ = 0.0;
        for (int j = threadIdx.x; j < m; j += 256) {
            double vr, vi; vof(j, vr, vi);
            if constexpr (CPLX) { const double cr = xc[2 * j], ci = xc[2 * j + 1]; ar_ += cr * vr + ci * vi; ai_ += cr * vi - ci * vr; }
            else ar_ += xc[j] * vr;
        }
        ar_ = tsum(ar_, sm);
        if (CPLX) ai_ = tsum(ai_, sm);
        if (threadIdx.x == 0) { coef[2 * b] = ar_; coef[2 * b + 1] = ai_; }
        return;
    }
    const int j0 = blockIdx.y * tcw;
    const int jn = min(tcw, m - j0);
    if ((int)threadIdx.x < jn) { double vr, vi; vof(j0 + threadIdx.x, vr, vi); vs[2 * threadIdx.x] = vr; vs[2 * threadIdx.x + 1] = vi; }
    __syncthreads();
    const int i = blockIdx.x * TRB + threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        scal[0] = tr; scal[1] = ti; scal[2] = beta; scal[3] = sr; scal[4] = si;
        tau[k * E] = tr; if (CPLX) tau[k * E + 1] = ti;
        e[k] = beta;
    }
    if (i >= m) return;
    if (blockIdx.y == 0) {                                               // v into the panel buffer (rows k+1 ..)
        double vr, vi; vof(i, vr, vi);
        X[((long)jj * n + k + 1 + i) * E] = vr; if (CPLX) X[((long)jj * n + k + 1 + i) * E + 1] = vi;
    }
    const double* a = A + ((long)(k + 1 + j0) * lda + (k + 1) + i) * E;
    double pr = 0.0, pi = 0.0;
    for (int j = 0; j < jn; ++j) {
        if constexpr (CPLX) {
            const double xr = a[0], xi = a[1], vr = vs[2 * j], vi = vs[2 * j + 1];
            pr += xr * vr - xi * vi; pi += xr * vi + xi * vr;
        } else pr += a[0] * vs[2 * j];
        a += lda * E;
    }
    double* o = part + ((long)blockIdx.y * m + i) * E;
    o[0] = pr; if (CPLX) o[1] = pi;
}

template <bool CPLX>
__global__ __launch_bounds__(256) void ltrd_w_kernel(double* __restrict__ A, long lda, int n, int k, int jj,
                                                     double* __restrict__ X, const double* __restrict__ part, int nch,
                                                     const double* __restrict__ coef, const double* __restrict__ scal,
                                                     double* __restrict__ dots)
{
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double cf[4 * LNB], sm[4];
    const int m = n - k - 1;
    if ((int)threadIdx.x < 4 * jj) cf[threadIdx.x] = coef[threadIdx.x];
    __syncthreads();
    const double tr = scal[0], ti = scal[1], beta = scal[2];
    const int i = blockIdx.x * TRB + threadIdx.x;
    double dr = 0.0, di = 0.0;
    if (i < m) {
        const long r = (long)k + 1 + i;
        double sr = 0.0, si = 0.0;
        for (int c = 0; c < nch; ++c) {
            const double* q = part + ((long)c * m + i) * E;
            sr += q[0]; if (CPLX) si += q[1];
        }
        const double* V = X;
        const double* W = X + (long)LNB * n * E;
        for (int c = 0; c < jj; ++c) {                  // p -= V[r,c] (W^H v)[c] + W[r,c] (V^H v)[c]
            const double vr = V[((long)c * n + r) * E], vi = CPLX ? V[((long)c * n + r) * E + 1] : 0.0;
            const double wr = W[((long)c * n + r) * E], wi = CPLX ? W[((long)c * n + r) * E + 1] : 0.0;
            const double cvr = cf[2 * c], cvi = cf[2 * c + 1];                      // V^H v
            const double cwr = cf[2 * (jj + c)], cwi = cf[2 * (jj + c) + 1];        // W^H v
            sr -= (vr * cwr - vi * cwi) + (wr * cvr - wi * cvi);
            si -= (vr * cwi + vi * cwr) + (wr * cvi + wi * cvr);
        }
        const double wr_ = tr * sr - ti * si, wi_ = tr * si + ti * sr;              // w = tau p (before the alpha v term)
        X[((long)(LNB + jj) * n + r) * E] = wr_; if (CPLX) X[((long)(LNB + jj) * n + r) * E + 1] = wi_;
        const double vr = X[((long)jj * n + r) * E], vi = CPLX ? X[((long)jj * n + r) * E + 1] : 0.0;
        dr = wr_ * vr + wi_ * vi;                       // conj(w) v
        di = wr_ * vi - wi_ * vr;
        // reflector store for the back-transformation: beta on the sub-diagonal, v[1:] below it
        if (i == 0) { A[((long)k * lda + r) * E] = beta; if (CPLX) A[((long)k * lda + r) * E + 1] = 0.0; }
        else { A[((long)k * lda + r) * E] = vr; if (CPLX) A[((long)k * lda + r) * E + 1] = vi; }
    }
    dr = tsum(dr, sm);
    if (CPLX) di = tsum(di, sm);
    if (threadIdx.x == 0) { dots[blockIdx.x * 2] = dr; dots[blockIdx.x * 2 + 1] = di; }
}

// end of a panel of jb columns (last column k_last): finalise its w, and YH (2 jb x m2, ld 2 jb) = [W | V]^H on the rows
// q0 .. n-1 of the trailing block, so that A22 -= [V | W] YH is one GEMM
template <bool CPLX>
__global__ __launch_bounds__(256) void ltrd_panel_end_kernel(double* __restrict__ X, int n, int q0, int jb, int k_last,
                                                             const double* __restrict__ tau, const double* __restrict__ dots,
                                                             int ndots, double* __restrict__ YH)
{
    constexpr int E = CPLX ? 2 : 1;
    double qr = 0.0, qi = 0.0;
    for (int b = 0; b < ndots; ++b) { qr += dots[2 * b]; qi += dots[2 * b + 1]; }
    const double tr = tau[k_last * E], ti = CPLX ? tau[k_last * E + 1] : 0.0;
    const double alr = -0.5 * (tr * qr - ti * qi), ali = -0.5 * (tr * qi + ti * qr);
    const int m2 = n - q0;
    double* V = X;
    double* W = X + (long)LNB * n * E;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < (long)m2 * jb; t += (long)gridDim.x * 256) {
        const int i = (int)(t % m2), c = (int)(t / m2);
        const long r = (long)q0 + i;
        const double vr = V[((long)c * n + r) * E], vi = CPLX ? V[((long)c * n + r) * E + 1] : 0.0;
        double wr = W[((long)c * n + r) * E], wi = CPLX ? W[((long)c * n + r) * E + 1] : 0.0;
        if (c == jb - 1) {
            wr += alr * vr - ali * vi; wi += alr * vi + ali * vr;
            W[((long)c * n + r) * E] = wr; if (CPLX) W[((long)c * n + r) * E + 1] = wi;
        }
        double* yw = YH + ((long)i * 2 * jb + c) * E;            // row c      : conj(W[r, c])
        double* yv = YH + ((long)i * 2 * jb + jb + c) * E;       // row jb + c : conj(V[r, c])
        yw[0] = wr; yv[0] = vr;
        if (CPLX) { yw[1] = -wi; yv[1] = -vi; }
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
        if (e_ != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e_, #x); }                             \
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
    // blocked reduction: panel buffer X = [V | W] (n x 2 LNB), YH (2 LNB x n), coefficients, scalars, partial norms
    const size_t szX = (size_t)2 * LNB * n * E, szY = szX, szSmall = (size_t)4 * LNB + 16 + (size_t)nrb_max + 8;
    // context-owned, grow-only block (round 3: hipMalloc + hipFree - a device-wide synchronisation - in every call)
    rc = c->ensure_buf(chase_hip_ctx::BUF_EIG, (2 * szv + szpart + 2 * nrb_max + 2 * (size_t)n + szv + szZ + szZ * E + 64 + szX + szY + szSmall) * sizeof(double));
    if (rc) return rc;
    double* blk = (double*)c->bufs[chase_hip_ctx::BUF_EIG];
    double* vbuf = blk; double* pbuf = vbuf + szv; double* part = pbuf + szv; double* dots = part + szpart;
    double* dd = dots + 2 * nrb_max; double* de = dd + n; double* tau = de + n; double* Zr = tau + szv; double* Zc = Zr + szZ;
    double* Xp = Zc + szZ * E + 64; double* YH = Xp + szX; double* coef = YH + szY; double* scal = coef + 4 * LNB;
    double* npart = scal + 16;
    const auto t_start = std::chrono::steady_clock::now();
    auto body = [&]() -> int {
        // symmetrise from the lower triangle so that the full-storage GEMV sees an exactly Hermitian matrix
        {
            int e = mirror_lower(st, A, lda, n, E);
            if (e) return hip_fail((hipError_t)e, "mirror_lower");
        }
        static const bool unblocked = getenv("CHASE_HIP_TRD_UNBLOCKED") != nullptr;
        if (unblocked) {
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
        } else {
            // blocked (xLATRD-style) reduction: see the ltrd_* kernels
            for (int p0 = 0; p0 < n - 1; p0 += LNB) {
                const int jb = (n - 1 - p0 < LNB) ? n - 1 - p0 : LNB;
                HC(hipMemsetAsync(Xp, 0, (size_t)2 * LNB * n * E * sizeof(double), st));
                int ndots_prev = 0;
                for (int jj = 0; jj < jb; ++jj) {
                    const int k = p0 + jj, m = n - k - 1;
                    const int nrk = (n - k + TRB - 1) / TRB, nrb = (m + TRB - 1) / TRB, tcw = (m >= 1792) ? LTCW : LTCW / 2, nch = (m + tcw - 1) / tcw;
                    const int nex = (2 * jj + nrb - 1) / nrb;             // extra grid rows: the 2 jj coefficient workgroups
                    if (cplx) {
                        HK(hipLaunchKernelGGL(ltrd_colupd_kernel<true>, dim3(nrk), dim3(256), 0, st, A, lda, n, k, jj, Xp, tau, dots, ndots_prev, npart, dd));
                        HK(hipLaunchKernelGGL(ltrd_gemv_kernel<true>, dim3(nrb, nch + nex), dim3(256), 0, st, A, lda, n, k, jj, Xp, npart, nrk, part, scal, tau, de, nch, coef, tcw));
                        HK(hipLaunchKernelGGL(ltrd_w_kernel<true>, dim3(nrb), dim3(256), 0, st, A, lda, n, k, jj, Xp, part, nch, coef, scal, dots));
                    } else {
                        HK(hipLaunchKernelGGL(ltrd_colupd_kernel<false>, dim3(nrk), dim3(256), 0, st, A, lda, n, k, jj, Xp, tau, dots, ndots_prev, npart, dd));
                        HK(hipLaunchKernelGGL(ltrd_gemv_kernel<false>, dim3(nrb, nch + nex), dim3(256), 0, st, A, lda, n, k, jj, Xp, npart, nrk, part, scal, tau, de, nch, coef, tcw));
                        HK(hipLaunchKernelGGL(ltrd_w_kernel<false>, dim3(nrb), dim3(256), 0, st, A, lda, n, k, jj, Xp, part, nch, coef, scal, dots));
                    }
                    ndots_prev = nrb;
                }
                // trailing block A[q0:, q0:] -= V W^H + W V^H through the MFMA GEMM
                const int q0 = p0 + jb, m2 = n - q0, k_last = q0 - 1;
                if (m2 <= 0) continue;
                {
                    unsigned gx = (unsigned)(((long)m2 * jb + 255) / 256); if (gx > 2048) gx = 2048;
                    if (cplx) HK(hipLaunchKernelGGL(ltrd_panel_end_kernel<true>, dim3(gx), dim3(256), 0, st, Xp, n, q0, jb, k_last, tau, dots, ndots_prev, YH));
                    else      HK(hipLaunchKernelGGL(ltrd_panel_end_kernel<false>, dim3(gx), dim3(256), 0, st, Xp, n, q0, jb, k_last, tau, dots, ndots_prev, YH));
                }
                const double mone[2] = {-1.0, 0.0}, one[2] = {1.0, 0.0};
                double* C22 = A + ((long)q0 * lda + q0) * E;
                const int ph = c->phase;
                c->phase = 0;
                int gr;
                if (jb == LNB) {
                    gr = c->gemm(cplx, 'N', m2, m2, 2 * jb, mone, Xp + (size_t)q0 * E, n, YH, 2 * jb, one, C22, lda);
                } else {
                    gr = c->gemm(cplx, 'N', m2, m2, jb, mone, Xp + (size_t)q0 * E, n, YH, 2 * jb, one, C22, lda);
                    if (!gr) gr = c->gemm(cplx, 'N', m2, m2, jb, mone, Xp + ((size_t)LNB * n + q0) * E, n, YH + (size_t)jb * E, 2 * jb, one, C22, lda);
                }
                c->phase = ph;
                if (gr) return gr;
            }
            // last diagonal entry (its column has no reflector): d[n-1] from the up-to-date trailing 1 x 1 block
            if (cplx) HK(hipLaunchKernelGGL(ltrd_colupd_kernel<true>, dim3(1), dim3(256), 0, st, A, lda, n, n - 1, 0, Xp, tau, dots, 0, npart, dd));
            else      HK(hipLaunchKernelGGL(ltrd_colupd_kernel<false>, dim3(1), dim3(256), 0, st, A, lda, n, n - 1, 0, Xp, tau, dots, 0, npart, dd));
        }
        // tridiagonal eigenproblem: divide & conquer with its O(n^2) / O(n^3) parts on the device (stedc_gpu.hip); on the host
        // (LAPACK dstedc / dstemr, eigenvectors uploaded) below CHASE_HIP_STEDC_GPU_MIN rows (default 512; 0: always host)
        std::vector<double> hd(n), hee(n), hz;
        HC(hipMemcpyAsync(hd.data(), dd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
        HC(hipMemcpyAsync(hee.data(), de, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToHost, st));
        HC(hipStreamSynchronize(st));
        auto t0b = std::chrono::steady_clock::now();
        hee[n - 1] = 0.0;
        // divide & conquer: threaded GEMMs inside, eigenvectors orthogonal to ~eps (MRRR gives ~1e-13 at n = 2560)
        static const bool use_mrrr = getenv("CHASE_HIP_TRIDIAG_MRRR") != nullptr;
        static const bool dbg = getenv("CHASE_HIP_HEEVD_TIMING") != nullptr;
        static const int dc_min = [] { const char* e = getenv("CHASE_HIP_STEDC_GPU_MIN"); return e ? atoi(e) : 512; }();
        auto t1 = std::chrono::steady_clock::now();
        int r2;
        bool on_device = !use_mrrr && dc_min > 0 && n >= dc_min;
        if (on_device) {
            r2 = stedc_gpu(c, n, hd.data(), hee.data(), w_host, Zr, n);
            if (r2) {                                       // never observed; the host solver is the safety net, loudly
                fprintf(stderr, "chase_hip: device divide & conquer failed (%s); falling back to the host solver\n", chase_hip_last_error());
                on_device = false;
            }
        }
        if (!on_device) {
            hz.resize((size_t)n * n);
            r2 = use_mrrr ? host_stemr(n, hd.data(), hee.data(), w_host, hz.data(), n)
                          : host_stedc(n, hd.data(), hee.data(), w_host, hz.data(), n);
            if (!r2) HC(hipMemcpyAsync(Zr, hz.data(), szZ * sizeof(double), hipMemcpyHostToDevice, st));
        }
        auto t2 = std::chrono::steady_clock::now();
        if (r2) return r2;
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
            fprintf(stderr, "heevd_gpu n=%d: tridiag %.1f ms, tridiagonal eig %.1f ms, back-transform %.1f ms\n", n,
                    ms(t_start, t0b), ms(t1, t2), ms(t2, t3));
        }
        return 0;
    };
    rc = body();
    const hipError_t es = hipStreamSynchronize(st);        // a faulting kernel must not pass for success
    if (rc == 0 && es != hipSuccess) return hip_fail(es, "heevd: hipStreamSynchronize");
    return rc;
}
