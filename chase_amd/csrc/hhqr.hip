// hhqr.hip — Householder QR fallback on the device: V (m x n) <- first n columns of Q with V = Q R.
//
// Reference behaviour replaced: cpu::houseHoulderQR = geqrf + ungqr (linalg/internal/cpu/cholqr1.hpp:203-210;
// cusolverDnTgeqrf / Tgqr in linalg/internal/cuda/cholqr.hpp:549-556), taken when potrf fails or CholQR is disabled
// (Impl/chase_cpu/chase_cpu.hpp:650-776).  Structure: blocked compact-WY (panel width 32, like the reference's
// distributed variant, linalg/internal/mpi/householder_qr.hpp:772-1054):
//   panel:    per column one reflector-generation launch (wave-shuffle norm) + one launch applying it to the
//             remaining panel columns (one workgroup per column: dot, then update)
//   T factor: G = V^H V with the split-K MFMA GEMM, forward recurrence in one workgroup
//   trailing: C -= V T^H (V^H C)  — three MFMA GEMMs;   form Q: backward accumulation with the same GEMMs.
// All scalars (tau, beta) stay on the device: no host synchronisation inside the factorisation.

#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../../include/chase_hip.h"
#include "../../include/chase_hip_grid.h"
#include "ctx.h"
#include "kernels.h"

namespace chase_hip {

constexpr int HNB = 32;        // sub-block width of the aggregated back-transformation (hh_apply_q_left)
constexpr int HMAX = 48;       // capacity of the panel kernels: largest panel width of the QR drivers (larft_kernel keeps T in 36 KB of static LDS)

// Panel width of the Householder QR drivers: 32 like the reference's (Impl/pchase_cpu/pchase_cpu.hpp:590-596,
// Impl/pchase_gpu/pchase_gpu.hpp:1065, linalg/internal/mpi/householder_qr.hpp:47-59), overridden by the reference's own
// CHASE_HOUSEHOLDER_NB (its outer block width; CHASE_QR_OUTER_BLOCK_NB is read as a synonym like the reference's
// mpi_qr_block_nb_env does), clamped to what the panel kernels hold.  Read at every call like the reference does.
static int householder_nb()
{
    int nb = HNB;
    for (const char* name : {"CHASE_QR_OUTER_BLOCK_NB", "CHASE_HOUSEHOLDER_NB"})
        if (const char* e = getenv(name)) { const long v = strtol(e, nullptr, 10); if (v > 0) nb = (int)(v > HMAX ? HMAX : v); }
    return nb < 2 ? 2 : nb;
}

__device__ __forceinline__ double wsum(double v)
{
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double bsum256(double v, double* sm)   // result broadcast to all threads
{
    v = wsum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// LAPACK xLARFG on column j of A (rows j..m-1): A[j,j] <- beta, A[j+1:,j] <- v (v_j = 1 implicit), tau[j].
template <bool CPLX>
__global__ __launch_bounds__(256) void house_gen_kernel(double* __restrict__ A, long lda, int m, int j,
                                                        double* __restrict__ tau)
{
    __shared__ double sm[4];
    constexpr int E = CPLX ? 2 : 1;
    double* col = A + ((long)j * lda) * E;
    double s = 0.0;
    for (long i = (long)(j + 1) * E + threadIdx.x; i < (long)m * E; i += 256) { const double x = col[i]; s += x * x; }
    const double xn2 = bsum256(s, sm);
    const double ar = col[(long)j * E], ai = CPLX ? col[(long)j * E + 1] : 0.0;
    if (xn2 == 0.0 && ai == 0.0) {                    // H = I
        if (threadIdx.x == 0) { tau[j * E] = 0.0; if (CPLX) tau[j * E + 1] = 0.0; }
        return;
    }
    const double nrm = sqrt(ar * ar + ai * ai + xn2);
    const double beta = (ar >= 0.0) ? -nrm : nrm;
    // scale = 1 / (alpha - beta)
    const double dr = ar - beta, di = ai;
    const double den = dr * dr + di * di;
    const double sr = dr / den, si = -di / den;
    __syncthreads();
    for (long i = (long)(j + 1) + threadIdx.x; i < m; i += 256) {
        if constexpr (CPLX) {
            const double xr = col[2 * i], xi = col[2 * i + 1];
            col[2 * i] = xr * sr - xi * si;
            col[2 * i + 1] = xr * si + xi * sr;
        } else {
            col[i] *= sr;
        }
    }
    if (threadIdx.x == 0) {
        tau[j * E] = (beta - ar) / beta;
        if (CPLX) tau[j * E + 1] = -ai / beta;
        col[(long)j * E] = beta;
        if (CPLX) col[(long)j * E + 1] = 0.0;
    }
}

// apply H_j^H = I - conj(tau) v v^H to columns c0 .. c0+ncols-1 (one workgroup per column)
template <bool CPLX>
__global__ __launch_bounds__(256) void house_apply_kernel(double* __restrict__ A, long lda, int m, int j, int c0,
                                                          const double* __restrict__ tau)
{
    __shared__ double sm[4];
    constexpr int E = CPLX ? 2 : 1;
    const int c = c0 + blockIdx.x;
    const double* v = A + ((long)j * lda) * E;
    double* a = A + ((long)c * lda) * E;
    const double tr = tau[j * E], ti = CPLX ? tau[j * E + 1] : 0.0;
    if (tr == 0.0 && ti == 0.0) return;
    double wr = 0.0, wi = 0.0;                         // w = v^H a
    for (long i = (long)(j + 1) + threadIdx.x; i < m; i += 256) {
        if constexpr (CPLX) {
            const double vr = v[2 * i], vi = v[2 * i + 1], xr = a[2 * i], xi = a[2 * i + 1];
            wr += vr * xr + vi * xi;
            wi += vr * xi - vi * xr;
        } else {
            wr += v[i] * a[i];
        }
    }
    wr = bsum256(wr, sm);
    if (CPLX) wi = bsum256(wi, sm);
    wr += a[(long)j * E];
    if (CPLX) wi += a[(long)j * E + 1];
    // f = conj(tau) * w
    const double fr = tr * wr + ti * wi, fi = tr * wi - ti * wr;
    __syncthreads();
    for (long i = (long)(j + 1) + threadIdx.x; i < m; i += 256) {
        if constexpr (CPLX) {
            const double vr = v[2 * i], vi = v[2 * i + 1];
            a[2 * i] -= fr * vr - fi * vi;
            a[2 * i + 1] -= fr * vi + fi * vr;
        } else {
            a[i] -= fr * v[i];
        }
    }
    if (threadIdx.x == 0) {
        a[(long)j * E] -= fr;
        if (CPLX) a[(long)j * E + 1] -= fi;
    }
}

// Vb (rows x nb, ld = rows) <- unit lower trapezoid of A[j0:m, j0:j0+nb]
__global__ __launch_bounds__(256) void extract_v_kernel(const double* __restrict__ A, long lda, int m, int j0, int nb,
                                                        int ept, double* __restrict__ Vb)
{
    const int rows = m - j0;
    const int c = blockIdx.y;
    if (c >= nb) return;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
        for (int e = 0; e < ept; ++e) {
            double v;
            if (r < c) v = 0.0;
            else if (r == c) v = (e == 0) ? 1.0 : 0.0;
            else v = A[((long)(j0 + c) * lda + j0 + r) * ept + e];
            Vb[((long)c * rows + r) * ept + e] = v;
        }
    }
}

// LAPACK xLARFT (forward, columnwise) from G = V^H V:  T[i,i] = tau_i;  T[0:i,i] = -tau_i * T[0:i,0:i] * G[0:i,i]
template <bool CPLX>
__global__ __launch_bounds__(64) void larft_kernel(const double* __restrict__ G, int nb, const double* __restrict__ tau,
                                                   double* __restrict__ T)
{
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double t[HMAX * HMAX * 2];
    __shared__ double g[HMAX * 2];
    const int r = threadIdx.x;
    for (int e = r; e < HMAX * HMAX * E; e += 64) t[e] = 0.0;
    __syncthreads();
    for (int i = 0; i < nb; ++i) {
        const double tr = tau[i * E], ti = CPLX ? tau[i * E + 1] : 0.0;
        if (r < i) { g[r * E] = G[((long)i * nb + r) * E]; if (CPLX) g[r * E + 1] = G[((long)i * nb + r) * E + 1]; }
        __syncthreads();
        if (r < i) {                                   // row r of upper-triangular T[0:i,0:i] times g
            double sr = 0.0, si = 0.0;
            for (int l = r; l < i; ++l) {
                const double ar = t[(l * HMAX + r) * E], ai = CPLX ? t[(l * HMAX + r) * E + 1] : 0.0;
                const double br = g[l * E], bi = CPLX ? g[l * E + 1] : 0.0;
                sr += ar * br - ai * bi;
                si += ar * bi + ai * br;
            }
            t[(i * HMAX + r) * E] = -(tr * sr - ti * si);
            if (CPLX) t[(i * HMAX + r) * E + 1] = -(tr * si + ti * sr);
        }
        if (r == i) { t[(i * HMAX + i) * E] = tr; if (CPLX) t[(i * HMAX + i) * E + 1] = ti; }
        __syncthreads();
    }
    for (int e = r; e < HMAX * HMAX * E; e += 64) T[e] = t[e];
}

// Q (m x n, ld = m) <- first n columns of the identity
__global__ __launch_bounds__(256) void set_identity_kernel(double* __restrict__ Q, int m, int n, int ept)
{
    const int c = blockIdx.y;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < m; r += gridDim.x * 256)
        for (int e = 0; e < ept; ++e) Q[((long)c * m + r) * ept + e] = (r == c && e == 0) ? 1.0 : 0.0;
}

// ---- row-distributed variant (chase_hip_houseqr_dist) ---------------------------------------------------------------
// The m x n block is split by rows over the ranks of a group; rank r holds mloc rows whose position in the "stacked"
// order (rank 0's rows, then rank 1's, ...) starts at `off`.  Column j's pivot is stacked row j.
// partial: what one all-reduce must carry for column j of the current panel (nc = remaining panel columns):
//   buf[0]            sum |x_i|^2 over the rows below the pivot            (x = A[:, j])
//   buf[1..2]         alpha = x[pivot]                                      (pivot owner only)
//   buf[3+2(c-1)..]   sum conj(x_i) a_i over the rows below the pivot      (column c = 1..nc right of j)
//   buf[3+2nc+2(c-1)] a[pivot]                                              (pivot owner only)
template <bool CPLX>
__global__ __launch_bounds__(256) void dhh_partial_kernel(const double* __restrict__ A, long lda, int mloc, long off, int j,
                                                          int nc, double* __restrict__ buf)
{
    __shared__ double sm[4];
    constexpr int E = CPLX ? 2 : 1;
    const long lp = (long)j - off;                                 // local pivot row (may lie outside [0, mloc))
    const long lo = lp + 1 < 0 ? 0 : lp + 1;
    const bool owner = lp >= 0 && lp < mloc;
    const double* x = A + (long)j * lda * E;
    const int c = blockIdx.x;
    if (c == 0) {
        double s = 0.0;
        for (long i = lo * E + threadIdx.x; i < (long)mloc * E; i += 256) { const double v = x[i]; s += v * v; }
        s = bsum256(s, sm);
        if (threadIdx.x == 0) {
            buf[0] = s;
            buf[1] = owner ? x[lp * E] : 0.0;
            buf[2] = (owner && CPLX) ? x[lp * E + 1] : 0.0;
        }
        return;
    }
    const double* a = A + (long)(j + c) * lda * E;
    double wr = 0.0, wi = 0.0;
    for (long i = lo + threadIdx.x; i < mloc; i += 256) {
        if constexpr (CPLX) {
            const double vr = x[2 * i], vi = x[2 * i + 1], ar = a[2 * i], ai = a[2 * i + 1];
            wr += vr * ar + vi * ai;
            wi += vr * ai - vi * ar;
        } else wr += x[i] * a[i];
    }
    wr = bsum256(wr, sm);
    if (CPLX) wi = bsum256(wi, sm);
    if (threadIdx.x == 0) {
        buf[3 + 2 * (c - 1)] = wr;
        buf[3 + 2 * (c - 1) + 1] = wi;
        buf[3 + 2 * nc + 2 * (c - 1)] = owner ? a[lp * E] : 0.0;
        buf[3 + 2 * nc + 2 * (c - 1) + 1] = (owner && CPLX) ? a[lp * E + 1] : 0.0;
    }
}

// update from the all-reduced buffer (every rank derives the same beta / tau / scale from the same numbers):
//   workgroup 0: v = [0 .. 0, 1, scale * x_below] into column jj of the panel buffer Vb (mloc rows), tau[j]
//   workgroup c: a_c -= conj(tau) (v^H a_c) v   (LAPACK xLARFG + xLARF of H^H, as house_gen / house_apply above)
template <bool CPLX>
__global__ __launch_bounds__(256) void dhh_update_kernel(double* __restrict__ A, long lda, int mloc, long off, int j, int nc,
                                                         const double* __restrict__ buf, double* __restrict__ Vb, int jj,
                                                         double* __restrict__ tau)
{
    constexpr int E = CPLX ? 2 : 1;
    const long lp = (long)j - off;
    const long lo = lp + 1 < 0 ? 0 : lp + 1;
    const bool owner = lp >= 0 && lp < mloc;
    const double xn2 = buf[0], ar = buf[1], ai = CPLX ? buf[2] : 0.0;
    const bool ident = (xn2 == 0.0 && ai == 0.0);                  // H = I
    double beta = ar, tr = 0.0, ti = 0.0, sr = 0.0, si = 0.0;
    if (!ident) {
        const double nrm = sqrt(ar * ar + ai * ai + xn2);
        beta = (ar >= 0.0) ? -nrm : nrm;
        tr = (beta - ar) / beta;
        ti = -ai / beta;
        const double dr = ar - beta, di = ai, den = dr * dr + di * di;
        sr = dr / den; si = -di / den;                              // scale = 1 / (alpha - beta)
    }
    const double* x = A + (long)j * lda * E;
    const int c = blockIdx.x;
    if (c == 0) {
        double* v = Vb + (long)jj * mloc * E;
        for (long i = threadIdx.x; i < mloc; i += 256) {
            double vr = 0.0, vi = 0.0;
            if (i == lp) vr = 1.0;
            else if (i >= lo) {
                if constexpr (CPLX) { const double xr = x[2 * i], xi = x[2 * i + 1]; vr = xr * sr - xi * si; vi = xr * si + xi * sr; }
                else vr = x[i] * sr;
            }
            v[i * E] = vr;
            if (CPLX) v[i * E + 1] = vi;
        }
        if (threadIdx.x == 0) { tau[j * E] = tr; if (CPLX) tau[j * E + 1] = ti; }
        return;
    }
    if (ident) return;
    // w = v^H a = conj(scale) * (x_below^H a_below) + a[pivot];  f = conj(tau) * w
    const double dr_ = buf[3 + 2 * (c - 1)], di_ = buf[3 + 2 * (c - 1) + 1];
    const double pr = buf[3 + 2 * nc + 2 * (c - 1)], pi = buf[3 + 2 * nc + 2 * (c - 1) + 1];
    const double wr = (sr * dr_ + si * di_) + pr, wi = (sr * di_ - si * dr_) + pi;
    const double fr = tr * wr + ti * wi, fi = tr * wi - ti * wr;
    // g = f * scale: a_below -= g * x_below
    const double gr = fr * sr - fi * si, gi = fr * si + fi * sr;
    double* a = A + (long)(j + c) * lda * E;
    for (long i = lo + threadIdx.x; i < mloc; i += 256) {
        if constexpr (CPLX) {
            const double xr = x[2 * i], xi = x[2 * i + 1];
            a[2 * i] -= gr * xr - gi * xi;
            a[2 * i + 1] -= gr * xi + gi * xr;
        } else a[i] -= gr * x[i];
    }
    if (owner && threadIdx.x == 0) {
        a[lp * E] -= fr;
        if (CPLX) a[lp * E + 1] -= fi;
    }
}

// Q (mloc x n, ld = mloc) <- this rank's rows of the first n columns of the identity in stacked order
__global__ __launch_bounds__(256) void set_identity_stacked_kernel(double* __restrict__ Q, int mloc, long off, int n, int ept)
{
    const int c = blockIdx.y;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < mloc; r += gridDim.x * 256)
        for (int e = 0; e < ept; ++e) Q[((long)c * mloc + r) * ept + e] = (off + r == c && e == 0) ? 1.0 : 0.0;
}

} // namespace chase_hip

using namespace chase_hip;

#define KL(x)                                                                                                          \
    do {                                                                                                               \
        x;                                                                                                             \
        hipError_t e_ = hipGetLastError();                                                                             \
        if (e_ != hipSuccess) return hip_fail(e_, #x);                                                                 \
    } while (0)
#define RC(x)                                                                                                          \
    do {                                                                                                               \
        int r_ = (x);                                                                                                  \
        if (r_) return r_;                                                                                             \
    } while (0)
#define HIPCHK_RET(x)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) return hip_fail(e_, #x);                                                                 \
    } while (0)

static int g3(chase_hip_ctx* c, bool cplx, char op, int m, int n, int k, double ar, const double* A, long lda,
              const double* B, long ldb, double br, double* C, long ldc)
{
    if (m <= 0 || n <= 0) return 0;
    const double alpha[2] = {ar, 0.0}, beta[2] = {br, 0.0};
    const int ph = c->phase;
    c->phase = 0;
    const int rc = c->gemm(cplx, op, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
    c->phase = ph;
    return rc;
}

namespace chase_hip {
// T_ss of every HNB-wide sub-block s of an aggregated block (LAPACK xLARFT forward / columnwise inside the sub-block) from the
// block's Gram matrix G = V^H V (nbig x nbig, ld ldg): written into T (ld ldt) at (HNB s, HNB s).  One wave per sub-block.
template <bool CPLX>
__global__ __launch_bounds__(64) void larft_diag_kernel(const double* __restrict__ G, int ldg, int nbig,
                                                        const double* __restrict__ tau, double* __restrict__ T, int ldt)
{
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double t[HNB * HNB * 2];
    __shared__ double g[HNB * 2];
    const int r = threadIdx.x;
    const int s0 = blockIdx.x * HNB;
    const int nb = (nbig - s0 < HNB) ? nbig - s0 : HNB;
    for (int e = r; e < HNB * HNB * E; e += 64) t[e] = 0.0;
    __syncthreads();
    for (int i = 0; i < nb; ++i) {
        const double tr = tau[(s0 + i) * E], ti = CPLX ? tau[(s0 + i) * E + 1] : 0.0;
        if (r < i) {
            const double* gp = G + ((long)(s0 + i) * ldg + s0 + r) * E;
            g[r * E] = gp[0];
            if (CPLX) g[r * E + 1] = gp[1];
        }
        __syncthreads();
        if (r < i) {                                   // row r of upper-triangular T[0:i,0:i] times g
            double sr = 0.0, si = 0.0;
            for (int l = r; l < i; ++l) {
                const double ar = t[(l * HNB + r) * E], ai = CPLX ? t[(l * HNB + r) * E + 1] : 0.0;
                const double br = g[l * E], bi = CPLX ? g[l * E + 1] : 0.0;
                sr += ar * br - ai * bi;
                si += ar * bi + ai * br;
            }
            t[(i * HNB + r) * E] = -(tr * sr - ti * si);
            if (CPLX) t[(i * HNB + r) * E + 1] = -(tr * si + ti * sr);
        }
        if (r == i) { t[(i * HNB + i) * E] = tr; if (CPLX) t[(i * HNB + i) * E + 1] = ti; }
        __syncthreads();
    }
    for (int e = r; e < nb * HNB * E; e += 64) {
        const int col = e / (HNB * E), rem = e % (HNB * E), row = rem / E, c = rem % E;
        if (row < nb) T[((long)(s0 + col) * ldt + s0 + row) * E + c] = t[(col * HNB + row) * E + c];
    }
}

// Cm (m x ncols, ldc) <- Q * Cm with Q = H_0 H_1 ... H_{nref-1}, reflectors in QR storage (unit at (k,k), tail below) in
// Vstore (m rows, ldv), scalars tau on the device.  Blocked compact-WY, backward over blocks (LAPACK xUNMQR 'L','N').
// Round 4: the reflectors are aggregated into blocks of ANB = 256 (round 3: 32) so that the three products of a block -
// V^H C, T (V^H C), C -= V (..) - run with an inner dimension of 256 instead of 32 (the MFMA GEMM spends a launch of K = 32 in
// its prologue and epilogue); the block's T is assembled from the 32-wide diagonal blocks (larft_diag_kernel) by the
// recurrence T(0:j, J) = -T(0:j, 0:j) G(0:j, J) T(J, J), two small products per sub-block.  n = 2560 complex: 20.3 -> see
// profiles/r04_heevd.txt.  CHASE_HIP_APPLYQ_BLOCK overrides the block size (a multiple of 32).
int hh_apply_q_left(chase_hip_ctx* c, bool cplx, const double* Vstore, long ldv, int m, int nref, const double* tau,
                    double* Cm, long ldc, int ncols)
{
    if (m <= 0 || nref <= 0 || ncols <= 0) return 0;
    const int E = cplx ? 2 : 1;
    hipStream_t st = c->stream;
    static const int anb_env = [] { const char* e = getenv("CHASE_HIP_APPLYQ_BLOCK"); return e ? atoi(e) : 256; }();
    const int ANB = (anb_env < HNB) ? HNB : (anb_env / HNB) * HNB;
    const int npan = (nref + ANB - 1) / ANB;
    const size_t szV = (size_t)m * ANB * E, szW = (size_t)ANB * ncols * E, szG = (size_t)ANB * ANB * E, szY = (size_t)ANB * HNB * E;
    int rcb = c->ensure_buf(chase_hip_ctx::BUF_APPLYQ, (szV + 2 * szW + 2 * szG + szY) * sizeof(double));
    if (rcb) return rcb;
    double* blk = (double*)c->bufs[chase_hip_ctx::BUF_APPLYQ];
    double* Vb = blk; double* W1 = Vb + szV; double* W2 = W1 + szW; double* G = W2 + szW; double* T = G + szG; double* Y = T + szG;
    auto body = [&]() -> int {
        for (int p = npan - 1; p >= 0; --p) {
            const int j0 = p * ANB, nb = (nref - j0 < ANB) ? nref - j0 : ANB;
            const int rows = m - j0;
            unsigned gx = (unsigned)((rows + 1023) / 1024); if (gx > 64) gx = 64; if (gx < 1) gx = 1;
            KL(hipLaunchKernelGGL(extract_v_kernel, dim3(gx, nb), dim3(256), 0, st, Vstore, ldv, m, j0, nb, E, Vb));
            RC(g3(c, cplx, 'C', nb, nb, rows, 1.0, Vb, rows, Vb, rows, 0.0, G, nb));
            if (hipError_t me = hipMemsetAsync(T, 0, (size_t)nb * nb * E * sizeof(double), st); me != hipSuccess)
                return hip_fail(me, "apply_q: memset T");
            const int nsub = (nb + HNB - 1) / HNB;
            if (cplx) KL(hipLaunchKernelGGL(larft_diag_kernel<true>, dim3(nsub), dim3(64), 0, st, G, nb, nb, tau + (size_t)j0 * E, T, nb));
            else      KL(hipLaunchKernelGGL(larft_diag_kernel<false>, dim3(nsub), dim3(64), 0, st, G, nb, nb, tau + (size_t)j0 * E, T, nb));
            for (int sb = 1; sb < nsub; ++sb) {
                const int s0 = sb * HNB, w = (nb - s0 < HNB) ? nb - s0 : HNB;
                // Y = T(0:s0, 0:s0) G(0:s0, s0:s0+w);  T(0:s0, s0:s0+w) = -Y T(s0:s0+w, s0:s0+w)
                RC(g3(c, cplx, 'N', s0, w, s0, 1.0, T, nb, G + (size_t)s0 * nb * E, nb, 0.0, Y, s0));
                RC(g3(c, cplx, 'N', s0, w, w, -1.0, Y, s0, T + ((size_t)s0 * nb + s0) * E, nb, 0.0, T + (size_t)s0 * nb * E, nb));
            }
            double* Cs = Cm + (size_t)j0 * E;
            RC(g3(c, cplx, 'C', nb, ncols, rows, 1.0, Vb, rows, Cs, ldc, 0.0, W1, nb));
            RC(g3(c, cplx, 'N', nb, ncols, nb, 1.0, T, nb, W1, nb, 0.0, W2, nb));
            RC(g3(c, cplx, 'N', rows, ncols, nb, -1.0, Vb, rows, W2, nb, 1.0, Cs, ldc));
        }
        return 0;
    };
    const int rc = body();
    const hipError_t es = hipStreamSynchronize(st);        // a faulting kernel must not pass for success
    if (rc == 0 && es != hipSuccess) return hip_fail(es, "houseqr: hipStreamSynchronize");
    return rc;
}
} // namespace chase_hip

extern "C" int chase_hip_houseqr(chase_hip_ctx* c, int cplx_, int m, int n, void* V_, long ldv)
{
    if (!c || !V_) return set_error(CHASE_HIP_EINVAL, "houseqr: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("houseqr", m, n, 0, 0);
    if (m < n || n < 0 || ldv < m) return set_error(CHASE_HIP_EINVAL, "houseqr: need m >= n and ldv >= m");
    if (n == 0) return 0;
    const bool cplx = cplx_ != 0;
    const int E = cplx ? 2 : 1;
    double* A = (double*)V_;
    hipStream_t st = c->stream;
    const int PNB = householder_nb();
    const int npan = (n + PNB - 1) / PNB;
    // one scratch block: Q (m x n) | Vb (m x nb) | W1, W2 (nb x n) | G (nb x nb) | T (npan x nb x nb) | tau (n)
    const size_t szQ = (size_t)m * n * E, szV = (size_t)m * PNB * E, szW = (size_t)PNB * n * E;
    const size_t szG = (size_t)PNB * PNB * E, szT = (size_t)npan * HMAX * HMAX * E, szTau = (size_t)n * E;
    double* blk = nullptr;
    hipError_t he = hipMalloc((void**)&blk, (szQ + szV + 2 * szW + szG + szT + szTau) * sizeof(double));
    if (he != hipSuccess) return set_error(CHASE_HIP_ENOMEM, "houseqr: scratch allocation failed");
    double* Q = blk; double* Vb = Q + szQ; double* W1 = Vb + szV; double* W2 = W1 + szW; double* G = W2 + szW;
    double* T = G + szG; double* tau = T + szT;
    int rc = 0;
    auto body = [&]() -> int {
        for (int p = 0; p < npan; ++p) {
            const int j0 = p * PNB, nb = (n - j0 < PNB) ? n - j0 : PNB, pend = j0 + nb;
            for (int j = j0; j < pend; ++j) {
                if (cplx) KL(hipLaunchKernelGGL(house_gen_kernel<true>, dim3(1), dim3(256), 0, st, A, ldv, m, j, tau));
                else      KL(hipLaunchKernelGGL(house_gen_kernel<false>, dim3(1), dim3(256), 0, st, A, ldv, m, j, tau));
                const int nc = pend - j - 1;
                if (nc > 0) {
                    if (cplx) KL(hipLaunchKernelGGL(house_apply_kernel<true>, dim3(nc), dim3(256), 0, st, A, ldv, m, j, j + 1, tau));
                    else      KL(hipLaunchKernelGGL(house_apply_kernel<false>, dim3(nc), dim3(256), 0, st, A, ldv, m, j, j + 1, tau));
                }
            }
            const int rows = m - j0;
            unsigned gx = (unsigned)((rows + 1023) / 1024); if (gx > 64) gx = 64; if (gx < 1) gx = 1;
            KL(hipLaunchKernelGGL(extract_v_kernel, dim3(gx, nb), dim3(256), 0, st, A, ldv, m, j0, nb, E, Vb));
            RC(g3(c, cplx, 'C', nb, nb, rows, 1.0, Vb, rows, Vb, rows, 0.0, G, nb));
            double* Tp = T + (size_t)p * HMAX * HMAX * E;
            if (cplx) KL(hipLaunchKernelGGL(larft_kernel<true>, dim3(1), dim3(64), 0, st, G, nb, tau + (size_t)j0 * E, Tp));
            else      KL(hipLaunchKernelGGL(larft_kernel<false>, dim3(1), dim3(64), 0, st, G, nb, tau + (size_t)j0 * E, Tp));
            const int nt = n - pend;
            if (nt > 0) {                               // C -= V T^H (V^H C)
                double* Cm = A + ((long)pend * ldv + j0) * E;
                RC(g3(c, cplx, 'C', nb, nt, rows, 1.0, Vb, rows, Cm, ldv, 0.0, W1, nb));
                RC(g3(c, cplx, 'C', nb, nt, nb, 1.0, Tp, HMAX, W1, nb, 0.0, W2, nb));
                RC(g3(c, cplx, 'N', rows, nt, nb, -1.0, Vb, rows, W2, nb, 1.0, Cm, ldv));
            }
        }
        // ---- form Q = H_1 ... H_k I(:, 1:n) by backward accumulation ----
        {
            unsigned gx = (unsigned)((m + 1023) / 1024); if (gx > 64) gx = 64; if (gx < 1) gx = 1;
            KL(hipLaunchKernelGGL(set_identity_kernel, dim3(gx, n), dim3(256), 0, st, Q, m, n, E));
        }
        for (int p = npan - 1; p >= 0; --p) {
            const int j0 = p * PNB, nb = (n - j0 < PNB) ? n - j0 : PNB;
            const int rows = m - j0, nq = n - j0;
            unsigned gx = (unsigned)((rows + 1023) / 1024); if (gx > 64) gx = 64; if (gx < 1) gx = 1;
            KL(hipLaunchKernelGGL(extract_v_kernel, dim3(gx, nb), dim3(256), 0, st, A, ldv, m, j0, nb, E, Vb));
            double* Tp = T + (size_t)p * HMAX * HMAX * E;
            double* Qs = Q + ((long)j0 * m + j0) * E;                    // Q[j0:m, j0:n]
            RC(g3(c, cplx, 'C', nb, nq, rows, 1.0, Vb, rows, Qs, m, 0.0, W1, nb));
            RC(g3(c, cplx, 'N', nb, nq, nb, 1.0, Tp, HMAX, W1, nb, 0.0, W2, nb));
            RC(g3(c, cplx, 'N', rows, nq, nb, -1.0, Vb, rows, W2, nb, 1.0, Qs, m));
        }
        int e = copy2d(st, Q, (long)m * E, A, ldv * E, (long)m * E, n);
        if (e) return hip_fail((hipError_t)e, "houseqr copy-back");
        return 0;
    };
    rc = body();
    const hipError_t es = hipStreamSynchronize(st);        // a faulting kernel must not pass for success
    (void)hipFree(blk);
    if (rc == 0 && es != hipSuccess) return hip_fail(es, "houseqr: hipStreamSynchronize");
    return rc;
}


/* Householder QR of a ROW-DISTRIBUTED m x n block (m = sum of the ranks' mloc >= n): V_loc <- this rank's rows of the first
 * n columns of Q.  Replaces the reference's distributed Householder (linalg/internal/mpi/householder_qr.hpp:99-223 panel
 * factorisation, :772-1054 blocked compact-WY + form Q, :1057-1417 block-cyclic rows; NCCL twin nccl/householder_qr.hpp:2957).
 * Differences by design: the pivot of column j is row j of the STACKED order (rank 0's rows first; `row_offset` = number of
 * rows held by the lower-ranked members), which makes block and block-cyclic row layouts the same code - Q then spans the
 * same nested column spaces as the reference's, with columns that may differ by a phase; every column costs ONE fused
 * all-reduce (|x|^2, alpha, x^H A_panel and the pivot row together) instead of three, every panel one more (V^H [V | C]),
 * all scalars stay on the device.  No buffer is larger than the local block. */
extern "C" int chase_hip_houseqr_dist(chase_hip_ctx* c, chase_hip_grid* grid, int group, int cplx_, int mloc, int n, void* V_,
                                      long ldv, long row_offset)
{
    if (!c || !grid || (!V_ && mloc > 0)) return set_error(CHASE_HIP_EINVAL, "houseqr_dist: NULL argument");
    (void)hipSetDevice(c->device);
    if (c->oplog_on) c->oplog_add("houseqr_dist", mloc, n, group, 0);
    if (mloc < 0 || n < 0 || ldv < mloc || row_offset < 0) return set_error(CHASE_HIP_EINVAL, "houseqr_dist: bad shape");
    if (n == 0) return 0;
    const bool cplx = cplx_ != 0;
    const int E = cplx ? 2 : 1;
    double* A = (double*)V_;
    hipStream_t st = c->stream;
    const int PNB = householder_nb();
    const int npan = (n + PNB - 1) / PNB;
    const int ml = mloc > 0 ? mloc : 1;                              // keep leading dimensions valid on an empty rank
    // scratch: Q (mloc x n) | Vb (mloc x nb) | Wr (nb x (nb + n)) | W2 (nb x n) | T (npan x nb x nb) | tau (n) | buf
    const size_t szQ = (size_t)ml * n * E, szV = (size_t)ml * PNB * E, szWr = (size_t)PNB * (PNB + n) * E;
    const size_t szW2 = (size_t)PNB * n * E, szT = (size_t)npan * HMAX * HMAX * E, szTau = (size_t)n * E;
    const size_t szBuf = 3 + 4 * (size_t)PNB + 8;
    double* blk = nullptr;
    if (hipMalloc((void**)&blk, (szQ + szV + szWr + szW2 + szT + szTau + szBuf) * sizeof(double)) != hipSuccess)
        return set_error(CHASE_HIP_ENOMEM, "houseqr_dist: scratch allocation failed");
    double* Q = blk; double* Vb = Q + szQ; double* Wr = Vb + szV; double* W2 = Wr + szWr; double* T = W2 + szW2;
    double* tau = T + szT; double* buf = tau + szTau;
    auto body = [&]() -> int {
        for (int p = 0; p < npan; ++p) {
            const int j0 = p * PNB, nb = (n - j0 < PNB) ? n - j0 : PNB, pend = j0 + nb;
            for (int j = j0; j < pend; ++j) {
                const int nc = pend - j - 1;
                if (cplx) KL(hipLaunchKernelGGL(dhh_partial_kernel<true>, dim3(1 + nc), dim3(256), 0, st, A, ldv, mloc, row_offset, j, nc, buf));
                else      KL(hipLaunchKernelGGL(dhh_partial_kernel<false>, dim3(1 + nc), dim3(256), 0, st, A, ldv, mloc, row_offset, j, nc, buf));
                RC(chase_hip_grid_allreduce(grid, group, buf, (size_t)(3 + 4 * nc), 0));
                if (cplx) KL(hipLaunchKernelGGL(dhh_update_kernel<true>, dim3(1 + nc), dim3(256), 0, st, A, ldv, mloc, row_offset, j, nc, buf, Vb, j - j0, tau));
                else      KL(hipLaunchKernelGGL(dhh_update_kernel<false>, dim3(1 + nc), dim3(256), 0, st, A, ldv, mloc, row_offset, j, nc, buf, Vb, j - j0, tau));
            }
            // one all-reduce for G = V^H V (-> T) and W = V^H C of the trailing columns
            const int nt = n - pend;
            RC(g3(c, cplx, 'C', nb, nb, mloc, 1.0, Vb, ml, Vb, ml, 0.0, Wr, nb));
            double* W1 = Wr + (size_t)nb * nb * E;
            if (nt > 0) RC(g3(c, cplx, 'C', nb, nt, mloc, 1.0, Vb, ml, A + (long)pend * ldv * E, ldv, 0.0, W1, nb));
            if (mloc == 0) HIPCHK_RET(hipMemsetAsync(Wr, 0, (size_t)nb * (nb + nt) * E * sizeof(double), st));
            RC(chase_hip_grid_allreduce(grid, group, Wr, (size_t)nb * (nb + nt) * E, 0));
            double* Tp = T + (size_t)p * HMAX * HMAX * E;
            if (cplx) KL(hipLaunchKernelGGL(larft_kernel<true>, dim3(1), dim3(64), 0, st, Wr, nb, tau + (size_t)j0 * E, Tp));
            else      KL(hipLaunchKernelGGL(larft_kernel<false>, dim3(1), dim3(64), 0, st, Wr, nb, tau + (size_t)j0 * E, Tp));
            if (nt > 0) {                                            // C -= V T^H (V^H C)
                RC(g3(c, cplx, 'C', nb, nt, nb, 1.0, Tp, HMAX, W1, nb, 0.0, W2, nb));
                RC(g3(c, cplx, 'N', mloc, nt, nb, -1.0, Vb, ml, W2, nb, 1.0, A + (long)pend * ldv * E, ldv));
            }
            // the factored columns are dead (R is not needed): they become the store of the panel's reflectors
            if (mloc > 0) {
                int e = copy2d(st, Vb, (long)ml * E, A + (long)j0 * ldv * E, ldv * E, (long)mloc * E, nb);
                if (e) return hip_fail((hipError_t)e, "houseqr_dist: reflector store");
            }
        }
        // ---- Q = H_0 ... H_{n-1} I(:, 0:n): backward accumulation, one all-reduce per panel ----
        {
            unsigned gx = (unsigned)((ml + 1023) / 1024); if (gx > 64) gx = 64; if (gx < 1) gx = 1;
            KL(hipLaunchKernelGGL(set_identity_stacked_kernel, dim3(gx, n), dim3(256), 0, st, Q, mloc, row_offset, n, E));
        }
        for (int p = npan - 1; p >= 0; --p) {
            const int j0 = p * PNB, nb = (n - j0 < PNB) ? n - j0 : PNB, nq = n - j0;
            const double* Vp = A + (long)j0 * ldv * E;               // stored reflectors (zeros above the pivots included)
            double* Qs = Q + (long)j0 * ml * E;
            RC(g3(c, cplx, 'C', nb, nq, mloc, 1.0, Vp, ldv, Qs, ml, 0.0, Wr, nb));
            if (mloc == 0) HIPCHK_RET(hipMemsetAsync(Wr, 0, (size_t)nb * nq * E * sizeof(double), st));
            RC(chase_hip_grid_allreduce(grid, group, Wr, (size_t)nb * nq * E, 0));
            double* Tp = T + (size_t)p * HMAX * HMAX * E;
            RC(g3(c, cplx, 'N', nb, nq, nb, 1.0, Tp, HMAX, Wr, nb, 0.0, W2, nb));
            RC(g3(c, cplx, 'N', mloc, nq, nb, -1.0, Vp, ldv, W2, nb, 1.0, Qs, ml));
        }
        if (mloc > 0) {
            int e = copy2d(st, Q, (long)ml * E, A, ldv * E, (long)mloc * E, n);
            if (e) return hip_fail((hipError_t)e, "houseqr_dist: copy-back");
        }
        return 0;
    };
    const int rc = body();
    const hipError_t es = hipStreamSynchronize(st);        // a faulting kernel must not pass for success
    (void)hipFree(blk);
    if (rc == 0 && es != hipSuccess) return hip_fail(es, "houseqr: hipStreamSynchronize");
    return rc;
}
