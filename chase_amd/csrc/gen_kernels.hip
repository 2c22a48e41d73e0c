// gen_kernels.hip — on-device input generators: counter-based normal RNG for the start vectors and
// the Clement test matrix, both addressed by GLOBAL indices so that any 2D block / block-cyclic shard of the data is
// generated in place, identically on every replica, without a host staging copy.
//
// Reference behaviour replaced: ChASEGPU::initVecs -> cuda::init_random_vectors (Philox4_32_10 via cuRAND,
// linalg/internal/cuda/random_normal_distribution.cu:21-95); the Clement generator of the solve tests/examples
// (tests/chase_serial_solve.cpp:52-66, examples/1_hello_world).  The GPU reference's random stream already differs from
// its CPU mt19937 stream (tests/chase_serial_solve_pseudo_bse_test.cpp:87-91), so only the distribution is contractual.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace chase_hip {

// Philox4x32-10 (Salmon et al., SC'11) — own implementation, no rocRAND
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4])
{
    #pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one Box-Muller pair from counter `ctr`
__device__ __forceinline__ void normal_pair(unsigned long long ctr, unsigned long long seed, double& z0, double& z1)
{
    unsigned o[4];
    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0x5eedu, 0u, (unsigned)seed, (unsigned)(seed >> 32), o);
    const double u1 = (((unsigned long long)o[0] << 21) ^ (o[1] >> 11)) * (1.0 / 9007199254740992.0) + (0.5 / 9007199254740992.0);
    const double u2 = (((unsigned long long)o[2] << 21) ^ (o[3] >> 11)) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    z0 = r * c; z1 = r * s;
}

// X[i, j] ~ N(0,1) (complex: re and im independent N(0,1)); element identity = global (row(i), gcol0 + j) with
// row(i) = grow0 + i (mb == 0: contiguous window) or grow0 + ((i / mb) * pr + pi) * mb + i % mb (block-cyclic rows)
template <bool CPLX>
__global__ __launch_bounds__(256) void fill_normal_kernel(double* __restrict__ X, long ldx, int m, int n, long grow0,
                                                          long gcol0, long gld, unsigned long long seed, int mb, int pr,
                                                          int pi)
{
    for (int j = blockIdx.y; j < n; j += gridDim.y) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
            const long gr = (mb > 0) ? grow0 + ((long)(i / mb) * pr + pi) * mb + i % mb : grow0 + i;
            const unsigned long long g = (unsigned long long)(gcol0 + j) * (unsigned long long)gld + (unsigned long long)gr;
            double z0, z1;
            if constexpr (CPLX) {
                normal_pair(g, seed, z0, z1);
                double* p = X + ((long)j * ldx + i) * 2;
                p[0] = z0; p[1] = z1;
            } else {
                normal_pair(g >> 1, seed, z0, z1);
                X[(long)j * ldx + i] = (g & 1) ? z1 : z0;
            }
        }
    }
}

// local shard of the Clement-type matrix of the reference's tests: H[g+1, g] = H[g, g+1] = sqrt(g (N + 1 - g)), else 0,
// plus (perturb != 0) a dense Hermitian perturbation perturb * N(0,1) on the entries 1 <= j < i < N exactly where
// tests/chase_serial_solve.cpp:68-90 adds one (drawn from Philox by global (min, max) index instead of mt19937(42)),
// everything multiplied by `scale`.
// local index -> global index: g = off + ((l / b) * p + q) * b + l % b   (block layout: b = local extent, p = 1, q = 0)
template <bool CPLX>
__global__ __launch_bounds__(256) void gen_clement_kernel(double* __restrict__ H, long ldh, int mloc, int nloc, long N,
                                                          int mb, int pr, int pi, long roff, int nb, int pc, int pj,
                                                          long coff, double scale, double perturb,
                                                          unsigned long long seed)
{
    for (int j = blockIdx.y; j < nloc; j += gridDim.y) {
        const long gj = coff + ((long)(j / nb) * pc + pj) * nb + j % nb;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < mloc; i += gridDim.x * 256) {
            const long gi = roff + ((long)(i / mb) * pr + pi) * mb + i % mb;
            double v = 0.0;
            const long d = gi - gj;
            if (d == 1 || d == -1) { const long g = (gi < gj) ? gi : gj; v = sqrt((double)g * (double)(N + 1 - g)); }
            double vi = 0.0;
            if (perturb != 0.0 && d != 0 && gi >= 1 && gj >= 1) {
                const long lo = (gi < gj) ? gi : gj, hi = (gi < gj) ? gj : gi;
                double z0, z1;
                normal_pair((unsigned long long)lo * (unsigned long long)N + (unsigned long long)hi, seed, z0, z1);
                v += perturb * z0;
                if (CPLX) vi = (gi < gj) ? perturb * z1 : -perturb * z1;     // upper triangle holds ep, lower conj(ep)
            }
            if constexpr (CPLX) { double* p = H + ((long)j * ldh + i) * 2; p[0] = scale * v; p[1] = scale * vi; }
            else H[(long)j * ldh + i] = scale * v;
        }
    }
}

// Synthetic Bethe-Salpeter matrix H = [[A, B], [-conj(B), -conj(A)]] of order N = 2h, any 2D shard (the reference's BSE
// benchmark reads such a matrix from a file, examples/5_bse_benchmark/5_bse_benchmark.cpp; its test fixture
// tests/linalg/internal/BSE_matrices/cdouble_random_BSE.bin has the same block structure):
//   A Hermitian:  A[i,i] = sqrt(dmin^2 + (dmax^2 - dmin^2) * i / (h - 1))  (eigenvalues of H^2, which the filter sees,
//                 spread uniformly),  A[i,j] = offdiag * (g0 + i g1) for i < j, conj below
//   B symmetric:  B[i,j] = B[j,i] = offdiag * (g2 + i g3)
// g ~ N(0,1) keyed on the unordered pair, so every shard sees the same global matrix.  S H = [[A, B], [conj(B), conj(A)]]
// is Hermitian, and positive definite when dmin > ~2 offdiag sqrt(N).  Real build: A, B real symmetric.
template <bool CPLX>
__global__ __launch_bounds__(256) void gen_bse_kernel(double* __restrict__ H, long ldh, int mloc, int nloc, long N, int mb,
                                                      int pr, int pi, int nb, int pc, int pj, double dmin, double dmax,
                                                      double offdiag, unsigned long long seed)
{
    const long h = N / 2;
    for (int j = blockIdx.y; j < nloc; j += gridDim.y) {
        const long gj = ((long)(j / nb) * pc + pj) * nb + j % nb;
        const bool bj = gj >= h;
        const long jj = bj ? gj - h : gj;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < mloc; i += gridDim.x * 256) {
            const long gi = ((long)(i / mb) * pr + pi) * mb + i % mb;
            const bool bi = gi >= h;
            const long ii = bi ? gi - h : gi;
            const long lo = ii < jj ? ii : jj, hi = ii < jj ? jj : ii;
            const unsigned long long key = (unsigned long long)lo * (unsigned long long)h + (unsigned long long)hi;
            double re, im = 0.0;
            if (bi == bj) {                                    // +-A (diagonal blocks)
                if (ii == jj) re = sqrt(dmin * dmin + (h > 1 ? (dmax * dmax - dmin * dmin) * (double)ii / (double)(h - 1) : 0.0));
                else {
                    double z0, z1;
                    normal_pair(key, seed, z0, z1);
                    re = offdiag * z0;
                    im = (ii < jj) ? offdiag * z1 : -offdiag * z1;
                }
                if (bi) re = -re;                              // -conj(a) = -re + i im
            } else {                                           // B (upper right) / -conj(B) (lower left)
                double z0, z1;
                normal_pair(key, seed ^ 0x9e3779b97f4a7c15ull, z0, z1);
                re = offdiag * z0;
                im = offdiag * z1;
                if (bi) re = -re;                              // -conj(b) = -re + i im
            }
            if constexpr (CPLX) { double* p = H + ((long)j * ldh + i) * 2; p[0] = re; p[1] = im; }
            else H[(long)j * ldh + i] = re;
        }
    }
}

static inline dim3 grid_for(int m, int n)
{
    unsigned gx = (unsigned)((m + 1023) / 1024); if (gx < 1) gx = 1; if (gx > 64) gx = 64;
    unsigned gy = n < 1 ? 1 : (n > 4096 ? 4096 : n);
    return dim3(gx, gy);
}

int fill_normal(hipStream_t st, bool cplx, double* X, long ldx, int m, int n, long grow0, long gcol0, long gld,
                unsigned long long seed, int mb, int pr, int pi)
{
    if (m <= 0 || n <= 0) return 0;
    if (cplx) hipLaunchKernelGGL(fill_normal_kernel<true>, grid_for(m, n), dim3(256), 0, st, X, ldx, m, n, grow0, gcol0, gld, seed, mb, pr, pi);
    else      hipLaunchKernelGGL(fill_normal_kernel<false>, grid_for(m, n), dim3(256), 0, st, X, ldx, m, n, grow0, gcol0, gld, seed, mb, pr, pi);
    return (int)hipGetLastError();
}

int gen_clement(hipStream_t st, bool cplx, double* H, long ldh, int mloc, int nloc, long N, int mb, int pr, int pi,
                long roff, int nb, int pc, int pj, long coff, double scale, double perturb, unsigned long long seed)
{
    if (mloc <= 0 || nloc <= 0) return 0;
    if (cplx) hipLaunchKernelGGL(gen_clement_kernel<true>, grid_for(mloc, nloc), dim3(256), 0, st, H, ldh, mloc, nloc, N, mb, pr, pi, roff, nb, pc, pj, coff, scale, perturb, seed);
    else      hipLaunchKernelGGL(gen_clement_kernel<false>, grid_for(mloc, nloc), dim3(256), 0, st, H, ldh, mloc, nloc, N, mb, pr, pi, roff, nb, pc, pj, coff, scale, perturb, seed);
    return (int)hipGetLastError();
}

int gen_bse(hipStream_t st, bool cplx, double* H, long ldh, int mloc, int nloc, long N, int mb, int pr, int pi, int nb,
            int pc, int pj, double dmin, double dmax, double offdiag, unsigned long long seed)
{
    if (mloc <= 0 || nloc <= 0) return 0;
    if (cplx) hipLaunchKernelGGL(gen_bse_kernel<true>, grid_for(mloc, nloc), dim3(256), 0, st, H, ldh, mloc, nloc, N, mb, pr, pi, nb, pc, pj, dmin, dmax, offdiag, seed);
    else      hipLaunchKernelGGL(gen_bse_kernel<false>, grid_for(mloc, nloc), dim3(256), 0, st, H, ldh, mloc, nloc, N, mb, pr, pi, nb, pc, pj, dmin, dmax, offdiag, seed);
    return (int)hipGetLastError();
}

} // namespace chase_hip
