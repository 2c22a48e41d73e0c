// stedc_gpu.hip — divide & conquer eigensolver for a real symmetric tridiagonal matrix with the O(n^2) and O(n^3) parts on the GPU.
//
// Replaces the host dstedc inside the projected-matrix eigensolver of the Rayleigh-Ritz step (reference: the whole HEEVD runs
// on the device, linalg/internal/nccl/rayleighRitz.hpp:170-173 -> cusolverDnXheevd; host twin cpu/rayleighRitz.hpp:104).
// Cuppen's divide & conquer with the Gu-Eisenstat stabilisation, the structure of LAPACK's dstedc / dlaed0-4:
//   * the matrix is torn into leaves (<= LEAF rows) by rank-one modifications; the leaves are solved on the host (tiny);
//   * pairs of blocks are merged level by level: T = Q (D + rho z z^T) Q^T, z = [last row of Q1, +-first row of Q2] / sqrt 2;
//       host (O(n) per merge): sort, deflation (negligible z entries, Givens rotations of nearly equal poles) - dlaed2's rules;
//       device: the rotations, the column gather, the k roots of the secular equation (one wave per root), the Gu-Eisenstat
//       vector z^ (one wave per entry), the eigenvector matrix S of the rank-one update (one workgroup per column) and the
//       product Q_new = Q S through the MFMA GEMM;
//   * two host synchronisations per level (z down, deflation lists up; new eigenvalues down).
// Secular equation: every root is bracketed between its two poles, the origin is moved to the nearer pole (dlaed4's trick: all
// differences d_i - lambda_j are formed as (d_i - d_origin) - mu_j and therefore to high relative accuracy) and mu is found by
// bisection on the BIT PATTERN of the double (61 steps reach the last ulp of mu wherever the root sits relative to the pole) -
// brute force that a GPU affords (k^2 * 62 divisions) in exchange for dlaed4's rational interpolation logic.  The Gu-Eisenstat
// recomputation of z from the computed roots then makes the eigenvectors numerically orthogonal whatever the roots' last bits
// are (prototype of exactly this scheme against random, Wilkinson, glued Wilkinson, graded, clustered and decoupled matrices:
// residual and orthogonality below 3 n eps).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>
#include "kernels.h"
#include "ctx.h"
#include "../../include/chase_hip.h"

namespace chase_hip {

int host_stedc(int n, double* d, double* e, double* w, double* Z, int ldz);

namespace {

constexpr int LEAF = 128;

__device__ __forceinline__ double wave_sum_all(double v)
{
    // butterfly: every lane ends with the same bits (a + b == b + a), fixed order
    #pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_prod_all(double v)
{
    #pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v *= __shfl_xor(v, m, 64);
    return v;
}

// z[t] = sgn[t] * Q[row[t], t] for all columns t of the level (row < 0: column not part of a merge)
__global__ __launch_bounds__(256) void dc_zgather_kernel(int n, const double* __restrict__ Q, long ldq, const int* __restrict__ row,
                                                         const double* __restrict__ sgn, double* __restrict__ z)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int r = row[t];
    z[t] = r >= 0 ? sgn[t] * Q[(long)t * ldq + r] : 0.0;
}

struct Rot { int a, b; double c, s; };
// Givens rotations of a merge applied to rows [r0, r0 + nr) of Q in list order (a row is independent of the others)
__global__ __launch_bounds__(256) void dc_rot_kernel(double* __restrict__ Q, long ldq, int r0, int nr, const Rot* __restrict__ rots, int nrot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nr) return;
    double* q = Q + r0 + i;
    for (int t = 0; t < nrot; ++t) {
        const Rot r = rots[t];
        const double qa = q[(long)r.a * ldq], qb = q[(long)r.b * ldq];
        q[(long)r.a * ldq] = r.c * qa + r.s * qb;
        q[(long)r.b * ldq] = -r.s * qa + r.c * qb;
    }
}

// one wave per root j of 1 + rho sum_i z_i^2 / (dl_i - lambda) = 0, dl ascending and distinct, all z_i != 0:
// lambda_j = dl[org_j] + mu_j with org_j the pole nearer to the root
__global__ __launch_bounds__(256) void dc_secular_kernel(int k, const double* __restrict__ dl, const double* __restrict__ z, double rho,
                                                         int* __restrict__ org, double* __restrict__ mu, double* __restrict__ lam)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= k) return;                                            // wave-uniform
    auto G = [&](int o, double shift) -> double {                  // secular function at lambda = dl[o] + shift
        const double dorg = dl[o];
        double s = 0.0;
        for (int i = lane; i < k; i += 64) {
            const double zi = z[i];
            s += zi * zi / ((dl[i] - dorg) - shift);
        }
        return 1.0 + rho * wave_sum_all(s);
    };
    int o = j;
    double sgn = 1.0, hi;
    if (j < k - 1) {
        const double half = 0.5 * (dl[j + 1] - dl[j]);
        if (!(G(j, half) >= 0.0)) { o = j + 1; sgn = -1.0; }       // root in the upper half of the interval: origin = upper pole
        hi = half;
    } else {
        double s = 0.0;
        for (int i = lane; i < k; i += 64) s += z[i] * z[i];
        hi = rho * wave_sum_all(s) * (1.0 + 8.0 * DBL_EPSILON) + DBL_MIN;   // lambda_max <= d_max + rho |z|^2
    }
    // bisection on the bit pattern of v = |mu| in [DBL_MIN, hi]: with the origin at the lower pole g grows with v (g < 0 next to
    // the pole), with the origin at the upper pole lambda = dl[o] - v and g falls with v (g > 0 next to the pole)
    long long lo_b = __double_as_longlong(DBL_MIN), hi_b = __double_as_longlong(hi);
    while (hi_b - lo_b > 1) {
        const long long mid = (lo_b + hi_b) >> 1;
        const double g = G(o, sgn * __longlong_as_double(mid));
        if ((g >= 0.0) == (sgn > 0.0)) hi_b = mid; else lo_b = mid;
    }
    const double vlo = __longlong_as_double(lo_b), vhi = __longlong_as_double(hi_b);
    const double glo = G(o, sgn * vlo), ghi = G(o, sgn * vhi);
    const double v = fabs(glo) < fabs(ghi) ? vlo : vhi;
    // a NaN secular function (garbage operands, a degenerate merge) would have steered the bisection silently - !(NaN >= 0)
    // reads as "negative": report it through the root, which the host checks for finiteness before it goes on
    const bool bad = (glo != glo) || (ghi != ghi);
    if (lane == 0) { org[j] = o; mu[j] = sgn * v; lam[j] = bad ? __longlong_as_double(0x7ff8000000000000ll) : dl[o] + sgn * v; }
}

// Gu-Eisenstat: z^_i = sign(z_i) sqrt( prod_j (lambda_j - dl_i) / prod_{j != i} (dl_j - dl_i) ), one wave per i
__global__ __launch_bounds__(256) void dc_zhat_kernel(int k, const double* __restrict__ dl, const double* __restrict__ z,
                                                      const int* __restrict__ org, const double* __restrict__ mu, double* __restrict__ zhat)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= k) return;
    const double di = dl[i];
    double p = 1.0;
    for (int j = lane; j < k; j += 64) {
        const double num = (dl[org[j]] - di) + mu[j];              // lambda_j - dl_i, formed from the shifted root
        p *= (j == i) ? num : num / (dl[j] - di);
    }
    p = wave_prod_all(p);
    if (lane == 0) zhat[i] = copysign(sqrt(fabs(p)), z[i]);
}

// column j of the eigenvector matrix of D + rho z^ z^^T: S_ij = z^_i / (dl_i - lambda_j), normalised; one workgroup per column
__global__ __launch_bounds__(256) void dc_vectors_kernel(int k, const double* __restrict__ dl, const double* __restrict__ zhat,
                                                         const int* __restrict__ org, const double* __restrict__ mu, double* __restrict__ S, long lds,
                                                         double* __restrict__ lam)
{
    __shared__ double part[4];
    const int j = blockIdx.x;
    const double dorg = dl[org[j]], m = mu[j];
    double* col = S + (long)j * lds;
    double s = 0.0;
    for (int i = threadIdx.x; i < k; i += 256) {
        const double t = zhat[i] / ((dl[i] - dorg) - m);
        col[i] = t;
        s += t * t;
    }
    s = wave_sum_all(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    const double inv = 1.0 / sqrt((part[0] + part[1]) + (part[2] + part[3]));
    for (int i = threadIdx.x; i < k; i += 256) col[i] *= inv;
    // a vector that cannot be normalised (non-finite or zero norm: garbage z^, a degenerate merge) must not reach the Ritz
    // vectors: poison its eigenvalue, which the host checks after this level (-> host solver)
    if (threadIdx.x == 0 && !(inv > 0.0 && inv <= 1.79769313486231570815e308)) lam[j] = __longlong_as_double(0x7ff8000000000000ll);
}

struct Merge {
    int o, n1, nm;          // block [o, o + nm) = [o, o + n1) + [o + n1, o + nm)
    double beta;            // coupling e[o + n1 - 1] (scaled)
    int k = 0;              // non-deflated eigenvalues
    double rho = 0;
    int rot0 = 0, nrot = 0; // slice of the level's rotation list
};

// error exits wait for the stream first: pageable host vectors that an asynchronous copy may still read go out of scope
// with the return (`st` is the stream variable of the enclosing function)
#define DCHK(x)                                                                                                        \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e_, #x); }                             \
    } while (0)
#define DCK(x)                                                                                                         \
    do {                                                                                                               \
        x;                                                                                                             \
        hipError_t e_ = hipGetLastError();                                                                             \
        if (e_ != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e_, #x); }                             \
    } while (0)

} // namespace

/* Eigen-decomposition of the symmetric tridiagonal matrix (d, e) of order n given on the HOST: eigenvalues ascending to w_host,
 * eigenvectors (n x n, real, column-major, leading dimension ldz) to the DEVICE array Z. */
int stedc_gpu(chase_hip_ctx* c, int n, const double* d_in, const double* e_in, double* w_host, double* Z, long ldz)
{
    if (!c || !d_in || !e_in || !w_host || !Z || n < 1 || ldz < n) return set_error(CHASE_HIP_EINVAL, "stedc_gpu: bad argument");
    hipStream_t st = c->stream;
    static const bool dbg = getenv("CHASE_HIP_HEEVD_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    // ---- scale to unit max norm (dstedc) ------------------------------------------------------------------------------
    double orgnrm = 0.0;
    for (int i = 0; i < n; ++i) orgnrm = std::max(orgnrm, std::fabs(d_in[i]));
    for (int i = 0; i + 1 < n; ++i) orgnrm = std::max(orgnrm, std::fabs(e_in[i]));
    std::vector<double> d(d_in, d_in + n), e(n, 0.0);
    if (orgnrm == 0.0 || !std::isfinite(orgnrm)) {
        if (!std::isfinite(orgnrm)) return set_error(CHASE_HIP_ENOTCONV, "stedc_gpu: non-finite tridiagonal matrix");
        orgnrm = 1.0;
    }
    for (int i = 0; i < n; ++i) d[i] /= orgnrm;
    for (int i = 0; i + 1 < n; ++i) e[i] = e_in[i] / orgnrm;
    // ---- leaves by repeated halving (dlaed0) -----------------------------------------------------------------------------
    std::vector<int> sizes{n};
    while (*std::max_element(sizes.begin(), sizes.end()) > LEAF) {
        std::vector<int> ns;
        for (int s : sizes) { ns.push_back(s / 2); ns.push_back(s - s / 2); }
        sizes.swap(ns);
    }
    const int nleaf = (int)sizes.size();
    std::vector<int> offs(nleaf + 1, 0);
    for (int i = 0; i < nleaf; ++i) offs[i + 1] = offs[i] + sizes[i];
    for (int i = 1; i < nleaf; ++i) {                              // rank-one tears
        const int b = offs[i];
        d[b - 1] -= std::fabs(e[b - 1]);
        d[b] -= std::fabs(e[b - 1]);
    }
    // ---- device scratch --------------------------------------------------------------------------------------------------
    const size_t nn = (size_t)n * n;
    int nlev = 0;
    for (int s = nleaf; s > 1; s >>= 1) ++nlev;
    const size_t dbl_count = 3 * nn + 8 * (size_t)n + (size_t)nlev * n + 64;
    const size_t int_count = 4 * (size_t)n + (size_t)nlev * n + 64;
    const size_t rot_count = (size_t)n + 8;
    // context-owned, grow-only (one hipMalloc for the life of the context instead of a hipMalloc + hipFree - a device-wide
    // synchronisation - in every Rayleigh-Ritz call; separate from heevd_gpu's block, which is alive around this call)
    {
        const int rcb = c->ensure_buf(chase_hip_ctx::BUF_STEDC, dbl_count * sizeof(double) + int_count * sizeof(int) + rot_count * sizeof(Rot));
        if (rcb) return rcb;
    }
    double* blk = (double*)c->bufs[chase_hip_ctx::BUF_STEDC];
    double* Q = blk; double* W = Q + nn; double* S = W + nn;
    double* zbuf = S + nn; double* dl = zbuf + n; double* zz = dl + n; double* mu = zz + n; double* lam = mu + n;
    double* zhat = lam + n; double* spare = zhat + n; double* zsgn_all = spare + 2 * (size_t)n;   // nlev * n
    Rot* rots_dev = (Rot*)(zsgn_all + (size_t)nlev * n + 8);
    int* perm_dev = (int*)(rots_dev + rot_count); int* dst_dev = perm_dev + n; int* org = dst_dev + n; int* ispare = org + n;
    int* zrow_all = ispare + n;                                                                  // nlev * n
    auto body = [&]() -> int {
        // ---- leaf problems on the host, eigenvector blocks onto the diagonal of Q ---------------------------------------
        DCHK(hipMemsetAsync(Q, 0, nn * sizeof(double), st));
        std::vector<double> D(n);
        size_t leaf_elems = 0;
        for (int s : sizes) leaf_elems += (size_t)s * s;
        std::vector<double> leafZ(leaf_elems);
        {
            size_t pos = 0;
            std::vector<double> dd(LEAF + 1), ee(LEAF + 1), ww(LEAF + 1);
            for (int i = 0; i < nleaf; ++i) {
                const int o = offs[i], s = sizes[i];
                double* Zl = leafZ.data() + pos;
                if (s == 1) { D[o] = d[o]; Zl[0] = 1.0; }
                else {
                    for (int t = 0; t < s; ++t) { dd[t] = d[o + t]; ee[t] = (t + 1 < s) ? e[o + t] : 0.0; }
                    const int rc = host_stedc(s, dd.data(), ee.data(), ww.data(), Zl, s);
                    if (rc) { (void)hipStreamSynchronize(st); return rc; }
                    for (int t = 0; t < s; ++t) D[o + t] = ww[t];
                }
                DCHK(hipMemcpy2DAsync(Q + (size_t)o * n + o, (size_t)n * sizeof(double), Zl, (size_t)s * sizeof(double),
                                      (size_t)s * sizeof(double), s, hipMemcpyHostToDevice, st));
                pos += (size_t)s * s;
            }
        }
        // ---- merge tree: which row of Q feeds z for every column, at every level (uploaded once) ------------------------
        std::vector<std::vector<Merge>> levels;
        {
            std::vector<std::pair<int, int>> blocks;                 // (offset, size)
            for (int i = 0; i < nleaf; ++i) blocks.emplace_back(offs[i], sizes[i]);
            while (blocks.size() > 1) {
                std::vector<Merge> lv;
                std::vector<std::pair<int, int>> nb;
                for (size_t i = 0; i + 1 < blocks.size(); i += 2) {
                    Merge m;
                    m.o = blocks[i].first; m.n1 = blocks[i].second; m.nm = blocks[i].second + blocks[i + 1].second;
                    m.beta = e[m.o + m.n1 - 1];
                    lv.push_back(m);
                    nb.emplace_back(m.o, m.nm);
                }
                levels.push_back(lv);
                blocks.swap(nb);
            }
        }
        {
            std::vector<int> zrow((size_t)nlev * n, -1);
            std::vector<double> zsg((size_t)nlev * n, 0.0);
            for (int l = 0; l < (int)levels.size(); ++l)
                for (const Merge& m : levels[l])
                    for (int t = 0; t < m.nm; ++t) {
                        zrow[(size_t)l * n + m.o + t] = (t < m.n1) ? m.o + m.n1 - 1 : m.o + m.n1;
                        zsg[(size_t)l * n + m.o + t] = (t < m.n1) ? 1.0 : (m.beta < 0.0 ? -1.0 : 1.0);
                    }
            if (nlev > 0) {
                DCHK(hipMemcpyAsync(zrow_all, zrow.data(), zrow.size() * sizeof(int), hipMemcpyHostToDevice, st));
                DCHK(hipMemcpyAsync(zsgn_all, zsg.data(), zsg.size() * sizeof(double), hipMemcpyHostToDevice, st));
                DCHK(hipStreamSynchronize(st));                      // the host vectors go out of scope
            }
        }
        const double eps = 0.5 * DBL_EPSILON;                       // dlamch('E')
        std::vector<double> z(n), h_dl(n), h_zz(n);
        std::vector<int> h_perm(n);
        std::vector<Rot> h_rots;
        std::vector<int> order;
        const int ph = c->phase;
        c->phase = 0;
        struct PhaseBack { chase_hip_ctx* c; int ph; ~PhaseBack() { c->phase = ph; } } phase_back{c, ph};
        for (int l = 0; l < (int)levels.size(); ++l) {
            auto& lv = levels[l];
            DCK(hipLaunchKernelGGL(dc_zgather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, Q, (long)n,
                                   zrow_all + (size_t)l * n, zsgn_all + (size_t)l * n, zbuf));
            DCHK(hipMemcpyAsync(z.data(), zbuf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
            DCHK(hipStreamSynchronize(st));
            // ---- host: deflation of every merge of the level (dlaed2) -----------------------------------------------------
            h_rots.clear();
            for (int t = 0; t < n; ++t) h_perm[t] = t;              // columns outside the merges stay where they are
            for (Merge& m : lv) {
                const int o = m.o, nm = m.nm;
                m.k = 0; m.rho = 2.0 * std::fabs(m.beta); m.rot0 = (int)h_rots.size(); m.nrot = 0;
                double zmax = 0.0, dmax = 0.0;
                for (int t = 0; t < nm; ++t) {
                    z[o + t] *= M_SQRT1_2;
                    zmax = std::max(zmax, std::fabs(z[o + t]));
                    dmax = std::max(dmax, std::fabs(D[o + t]));
                }
                const double tol = 8.0 * eps * std::max(dmax, zmax);
                if (m.rho * zmax <= tol) continue;                   // nothing couples: the blocks' eigenpairs stand
                order.resize(nm);
                std::iota(order.begin(), order.end(), o);
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return D[a] < D[b]; });
                std::vector<int> nd, defl;
                int pj = -1;
                for (int idx : order) {
                    if (m.rho * std::fabs(z[idx]) <= tol) { defl.push_back(idx); continue; }
                    if (pj < 0) { pj = idx; continue; }
                    double s = z[pj], cc = z[idx];
                    const double tau = std::hypot(cc, s), t = D[idx] - D[pj];
                    cc /= tau; s = -s / tau;
                    if (std::fabs(t * cc * s) <= tol) {              // two poles numerically equal: rotate one z entry away
                        z[idx] = tau; z[pj] = 0.0;
                        h_rots.push_back(Rot{pj, idx, cc, s});
                        const double tt = D[pj] * cc * cc + D[idx] * s * s;
                        D[idx] = D[pj] * s * s + D[idx] * cc * cc;
                        D[pj] = tt;
                        defl.push_back(pj);
                        pj = idx;
                    } else {
                        nd.push_back(pj);
                        pj = idx;
                    }
                }
                if (pj >= 0) nd.push_back(pj);
                if (nd.empty()) continue;                            // every z entry negligible (no rotation happened): as above
                m.k = (int)nd.size();
                m.nrot = (int)h_rots.size() - m.rot0;
                // new physical order of the block: [non-deflated, ascending poles | deflated]
                std::vector<double> dnew(nm);
                for (int t = 0; t < m.k; ++t) { h_perm[o + t] = nd[t]; h_dl[o + t] = D[nd[t]]; h_zz[o + t] = z[nd[t]]; }
                for (int t = 0; t < (int)defl.size(); ++t) { h_perm[o + m.k + t] = defl[t]; dnew[m.k + t] = D[defl[t]]; }
                for (int t = m.k; t < nm; ++t) D[o + t] = dnew[t];   // D[o .. o+k) comes back from the device below
            }
            if (h_rots.size() > rot_count) { (void)hipStreamSynchronize(st); return set_error(CHASE_HIP_ENOTCONV, "stedc_gpu: rotation list overflow"); }
            DCHK(hipMemcpyAsync(perm_dev, h_perm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
            DCHK(hipMemcpyAsync(dl, h_dl.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
            DCHK(hipMemcpyAsync(zz, h_zz.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
            if (!h_rots.empty())
                DCHK(hipMemcpyAsync(rots_dev, h_rots.data(), h_rots.size() * sizeof(Rot), hipMemcpyHostToDevice, st));
            // ---- device: rotations, gather, secular equation, z^, S, Q S --------------------------------------------------
            bool any = false;
            for (const Merge& m : lv) {
                if (m.k == 0) continue;
                any = true;
                const int o = m.o, nm = m.nm, k = m.k;
                if (m.nrot)
                    DCK(hipLaunchKernelGGL(dc_rot_kernel, dim3((nm + 255) / 256), dim3(256), 0, st, Q, (long)n, o, nm, rots_dev + m.rot0, m.nrot));
                // W[o:o+nm, o+t] = Q[o:o+nm, perm[o+t]]
                {
                    int e2 = copy_cols_indexed_range(st, Q + o, (long)n, W + o, (long)n, (long)nm, perm_dev + o, o, nm);
                    if (e2) { (void)hipStreamSynchronize(st); return hip_fail((hipError_t)e2, "stedc_gpu gather"); }
                }
                DCK(hipLaunchKernelGGL(dc_secular_kernel, dim3((k + 3) / 4), dim3(256), 0, st, k, dl + o, zz + o, m.rho, org + o, mu + o, lam + o));
                DCK(hipLaunchKernelGGL(dc_zhat_kernel, dim3((k + 3) / 4), dim3(256), 0, st, k, dl + o, zz + o, org + o, mu + o, zhat + o));
                DCK(hipLaunchKernelGGL(dc_vectors_kernel, dim3(k), dim3(256), 0, st, k, dl + o, zhat + o, org + o, mu + o, S + (size_t)o * n + o, (long)n, lam + o));
                const double one[2] = {1.0, 0.0}, zero[2] = {0.0, 0.0};
                int gr = c->gemm(false, 'N', nm, k, k, one, W + (size_t)o * n + o, n, S + (size_t)o * n + o, n, zero, Q + (size_t)o * n + o, n);
                if (gr) { (void)hipStreamSynchronize(st); return gr; }
                if (k < nm) {
                    int e3 = copy2d(st, W + (size_t)(o + k) * n + o, (long)n, Q + (size_t)(o + k) * n + o, (long)n, (long)nm, nm - k);
                    if (e3) { (void)hipStreamSynchronize(st); return hip_fail((hipError_t)e3, "stedc_gpu deflated copy"); }
                }
            }
            if (any) DCHK(hipMemcpyAsync(h_dl.data(), lam, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
            DCHK(hipStreamSynchronize(st));                          // also: the host lists above are reused by the next level
            if (any) {
                for (const Merge& m : lv)
                    for (int t = 0; t < m.k; ++t) {
                        if (!std::isfinite(h_dl[m.o + t])) return set_error(CHASE_HIP_ENOTCONV, "stedc_gpu: secular equation produced a non-finite root");
                        D[m.o + t] = h_dl[m.o + t];
                    }
            }
        }
        // ---- ascending order ---------------------------------------------------------------------------------------------
        std::vector<int> p(n);
        std::iota(p.begin(), p.end(), 0);
        std::stable_sort(p.begin(), p.end(), [&](int a, int b) { return D[a] < D[b]; });
        for (int t = 0; t < n; ++t) w_host[t] = D[p[t]] * orgnrm;
        DCHK(hipMemcpyAsync(perm_dev, p.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
        int e4 = copy_cols_indexed_range(st, Q, (long)n, Z, ldz, (long)n, perm_dev, 0, n);
        if (e4) { (void)hipStreamSynchronize(st); return hip_fail((hipError_t)e4, "stedc_gpu final order"); }
        DCHK(hipStreamSynchronize(st));
        return 0;
    };
    int rc = body();
    const hipError_t es = hipStreamSynchronize(st);        // a faulting kernel must not pass for success
    if (rc == 0 && es != hipSuccess) rc = hip_fail(es, "stedc_gpu: hipStreamSynchronize");
    if (dbg && rc == 0)
        fprintf(stderr, "stedc_gpu n=%d: %d leaves, %.2f ms\n", n, nleaf,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    return rc;
}

} // namespace chase_hip
