// host_eig_builtin.cpp — self-contained small dense eigensolvers, used when no host LAPACK can be bound at run time.
//
// The hot path keeps two small problems on the host (north star: "small HEEV on host"): the Hermitian eigenproblem of the
// projected matrix (reference: lapackpp::t_heevd, linalg/internal/cpu/rayleighRitz.hpp:104) and the symmetric tridiagonal
// eigenproblems of Lanczos / of the GPU tridiagonalisation (reference: t_stemr, linalg/internal/cpu/lanczos.hpp:188).  The
// reference links a LAPACK at build time; this library binds one with dlopen (host_lapack.cpp) and, so that a plain C
// application on a box without MKL / OpenBLAS still works, carries these own implementations as the last provider
// ("builtin"):
//   * tridiagonal: implicit-shift QL iteration with Wilkinson shifts, eigenvectors accumulated (rotations of one sweep are
//     recorded and applied to row blocks of Z by a few threads for larger n);
//   * Hermitian / symmetric dense: Householder reduction to a complex-Hermitian tridiagonal matrix by Hermitian reflectors
//     P = I - beta u u^H, a unitary diagonal scaling that makes it real, the QL iteration above, back-transformation;
//   * Cholesky factorisation and triangular solves for the dense core of the pseudo-Hermitian Rayleigh-Ritz.
// Accuracy is that of the textbook algorithms (backward stable); speed is secondary (n is a few hundred here; with a GPU
// present the O(n^3) parts of larger problems run on the device, hetrd.hip).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>
#include "host_lapack.h"

namespace chase_hip {
namespace {

using cd = std::complex<double>;

// applies the recorded rotations (c[i], s[i]) for i = hi-1 down to lo to columns (i, i+1) of Z, rows [r0, r1)
void apply_sweep(double* Z, int ldz, int r0, int r1, int lo, int hi, const double* c, const double* s)
{
    for (int i = hi - 1; i >= lo; --i) {
        double* zi = Z + (size_t)i * ldz;
        double* zj = zi + ldz;
        const double ci = c[i], si = s[i];
        for (int k = r0; k < r1; ++k) {
            const double f = zj[k];
            zj[k] = si * zi[k] + ci * f;
            zi[k] = ci * zi[k] - si * f;
        }
    }
}

} // namespace

// symmetric tridiagonal (d[0..n), e[0..n-1) sub-diagonal; e[n-1] is workspace): eigenvalues ascending in w, eigenvectors
// in the columns of Z (n x n, column-major).  d and e are destroyed.  Returns 0, or > 0 if an eigenvalue did not converge.
int builtin_tridiag_eig(int n, double* d, double* e, double* w, double* Z, int ldz)
{
    if (n <= 0) return 0;
    for (int j = 0; j < n; ++j) {
        std::memset(Z + (size_t)j * ldz, 0, sizeof(double) * (size_t)n);
        Z[j + (size_t)j * ldz] = 1.0;
    }
    e[n - 1] = 0.0;
    std::vector<double> rc((size_t)n), rs((size_t)n);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int nthreads = n >= 512 ? (int)std::min<unsigned>(16, hw) : 1;
    const double eps = std::numeric_limits<double>::epsilon();
    for (int l = 0; l < n; ++l) {
        int iter = 0;
        while (true) {
            int m = l;
            for (; m < n - 1; ++m) {
                const double dd = std::abs(d[m]) + std::abs(d[m + 1]);
                if (std::abs(e[m]) <= eps * dd) break;
            }
            if (m == l) break;
            if (++iter > 60) return l + 1;
            // Wilkinson shift from the leading 2 x 2 block
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = std::hypot(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::abs(r) : -std::abs(r)));
            double s = 1.0, c = 1.0, p = 0.0;
            int i = m - 1;
            bool deflated = false;
            for (; i >= l; --i) {
                double f = s * e[i];
                const double b = c * e[i];
                r = std::hypot(f, g);
                e[i + 1] = r;
                if (r == 0.0) {                       // underflow: skip the transformation, restart
                    d[i + 1] -= p;
                    e[m] = 0.0;
                    deflated = true;
                    break;
                }
                s = f / r;
                c = g / r;
                g = d[i + 1] - p;
                r = (d[i] - g) * s + 2.0 * c * b;
                p = s * r;
                d[i + 1] = g + p;
                g = c * r - b;
                rc[(size_t)i] = c;
                rs[(size_t)i] = s;
            }
            const int lo = deflated ? i + 1 : l;
            // eigenvector update: the sweep's rotations on columns (i, i+1), i = m-1 .. lo
            if (nthreads == 1 || m - lo < 8) {
                apply_sweep(Z, ldz, 0, n, lo, m, rc.data(), rs.data());
            } else {
                std::vector<std::thread> th;
                const int chunk = (n + nthreads - 1) / nthreads;
                for (int t = 0; t < nthreads; ++t) {
                    const int r0 = t * chunk, r1 = std::min(n, r0 + chunk);
                    if (r0 < r1) th.emplace_back(apply_sweep, Z, ldz, r0, r1, lo, m, rc.data(), rs.data());
                }
                for (auto& t : th) t.join();
            }
            if (deflated) continue;
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
        }
    }
    // ascending order
    std::vector<int> idx((size_t)n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] < d[b]; });
    std::vector<double> tmp((size_t)n * n);
    for (int j = 0; j < n; ++j) {
        w[j] = d[idx[(size_t)j]];
        std::memcpy(tmp.data() + (size_t)j * n, Z + (size_t)idx[(size_t)j] * ldz, sizeof(double) * (size_t)n);
    }
    for (int j = 0; j < n; ++j) std::memcpy(Z + (size_t)j * ldz, tmp.data() + (size_t)j * n, sizeof(double) * (size_t)n);
    return 0;
}

// A (n x n column-major, lda; real or interleaved complex) Hermitian, LOWER triangle referenced: eigenvalues ascending in
// w, orthonormal eigenvectors overwrite A.
int builtin_heevd(bool cplx, int n, double* A_, int lda, double* w)
{
    if (n <= 0) return 0;
    // work in complex arithmetic on a full Hermitian copy (the problems are small; the real case is the same code)
    std::vector<cd> A((size_t)n * n), Q((size_t)n * n, cd(0.0));
    auto in = [&](int i, int j) -> cd {
        return cplx ? cd(A_[2 * (i + (size_t)j * lda)], A_[2 * (i + (size_t)j * lda) + 1]) : cd(A_[i + (size_t)j * lda], 0.0);
    };
    for (int j = 0; j < n; ++j) {
        A[j + (size_t)j * n] = cd(in(j, j).real(), 0.0);
        for (int i = j + 1; i < n; ++i) {
            A[i + (size_t)j * n] = in(i, j);
            A[j + (size_t)i * n] = std::conj(in(i, j));
        }
        Q[j + (size_t)j * n] = 1.0;
    }
    std::vector<cd> u((size_t)n), p((size_t)n), sub((size_t)n, cd(0.0));
    for (int k = 0; k + 2 < n; ++k) {
        // reflector P = I - beta u u^H on rows/columns k+1..n-1 mapping x = A[k+1:, k] to sigma e_1
        const int m = n - k - 1;
        cd* x = &A[(k + 1) + (size_t)k * n];
        double xn2 = 0.0;
        for (int i = 1; i < m; ++i) xn2 += std::norm(x[i]);
        if (xn2 == 0.0) { sub[(size_t)k] = x[0]; continue; }
        const double nx = std::sqrt(std::norm(x[0]) + xn2);
        const cd phase = (std::abs(x[0]) == 0.0) ? cd(1.0) : x[0] / std::abs(x[0]);
        const cd sigma = -phase * nx;
        for (int i = 0; i < m; ++i) u[(size_t)i] = x[i];
        u[0] -= sigma;
        double un2 = 0.0;
        for (int i = 0; i < m; ++i) un2 += std::norm(u[(size_t)i]);
        const double beta = 2.0 / un2;
        // trailing block B = A[k+1:, k+1:]:  B <- P B P = B - u w^H - w u^H,  p = beta B u, w = p - (beta u^H p / 2) u
        for (int i = 0; i < m; ++i) p[(size_t)i] = 0.0;
        for (int j = 0; j < m; ++j) {
            const cd uj = u[(size_t)j];
            const cd* col = &A[(k + 1) + (size_t)(k + 1 + j) * n];
            for (int i = 0; i < m; ++i) p[(size_t)i] += col[i] * uj;
        }
        cd up = 0.0;
        for (int i = 0; i < m; ++i) { p[(size_t)i] *= beta; up += std::conj(u[(size_t)i]) * p[(size_t)i]; }
        const double gamma = 0.5 * beta * up.real();
        for (int i = 0; i < m; ++i) p[(size_t)i] -= gamma * u[(size_t)i];          // p is w now
        for (int j = 0; j < m; ++j) {
            const cd wj = std::conj(p[(size_t)j]), uj = std::conj(u[(size_t)j]);
            cd* col = &A[(k + 1) + (size_t)(k + 1 + j) * n];
            for (int i = 0; i < m; ++i) col[i] -= u[(size_t)i] * wj + p[(size_t)i] * uj;
        }
        sub[(size_t)k] = sigma;
        // Q <- Q P (columns k+1..n-1)
        for (int i = 0; i < n; ++i) {
            cd t = 0.0;
            for (int j = 0; j < m; ++j) t += Q[i + (size_t)(k + 1 + j) * n] * u[(size_t)j];
            t *= beta;
            for (int j = 0; j < m; ++j) Q[i + (size_t)(k + 1 + j) * n] -= t * std::conj(u[(size_t)j]);
        }
    }
    if (n >= 2) sub[(size_t)n - 2] = A[(n - 1) + (size_t)(n - 2) * n];
    // A = Q Tc Q^H with Tc Hermitian tridiagonal (sub-diagonal sub[k], complex).  Tc = D T D^H, D unitary diagonal with
    // D[0] = 1, D[k+1] = D[k] * sub[k] / |sub[k]|, T real symmetric with off-diagonal |sub[k]|.
    std::vector<double> d((size_t)n), e((size_t)n, 0.0), Z((size_t)n * n);
    std::vector<cd> D((size_t)n);
    D[0] = 1.0;
    for (int k = 0; k < n; ++k) d[(size_t)k] = A[k + (size_t)k * n].real();
    for (int k = 0; k + 1 < n; ++k) {
        const double a = std::abs(sub[(size_t)k]);
        e[(size_t)k] = a;
        D[(size_t)k + 1] = (a == 0.0) ? D[(size_t)k] : D[(size_t)k] * sub[(size_t)k] / a;
    }
    const int info = builtin_tridiag_eig(n, d.data(), e.data(), w, Z.data(), n);
    if (info) return info;
    // eigenvectors (Q D) Z
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Q[i + (size_t)j * n] *= D[(size_t)j];
    std::vector<cd> col((size_t)n);
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) col[(size_t)i] = 0.0;
        for (int k = 0; k < n; ++k) {
            const double z = Z[k + (size_t)j * n];
            const cd* q = &Q[(size_t)k * n];
            for (int i = 0; i < n; ++i) col[(size_t)i] += q[i] * z;
        }
        for (int i = 0; i < n; ++i) {
            if (cplx) { A_[2 * (i + (size_t)j * lda)] = col[(size_t)i].real(); A_[2 * (i + (size_t)j * lda) + 1] = col[(size_t)i].imag(); }
            else A_[i + (size_t)j * lda] = col[(size_t)i].real();
        }
    }
    return 0;
}

// A = L L^H in place (lower triangle); returns LAPACK-style info (> 0: leading minor not positive definite)
int builtin_potrf_lower(bool cplx, int n, double* A_, int lda)
{
    const int E = cplx ? 2 : 1;
    auto at = [&](int i, int j) -> double* { return A_ + E * (i + (size_t)j * lda); };
    for (int j = 0; j < n; ++j) {
        double s = at(j, j)[0];
        for (int k = 0; k < j; ++k) s -= at(j, k)[0] * at(j, k)[0] + (cplx ? at(j, k)[1] * at(j, k)[1] : 0.0);
        if (!(s > 0.0)) return j + 1;
        const double ljj = std::sqrt(s);
        at(j, j)[0] = ljj;
        if (cplx) at(j, j)[1] = 0.0;
        for (int i = j + 1; i < n; ++i) {
            cd v = cplx ? cd(at(i, j)[0], at(i, j)[1]) : cd(at(i, j)[0], 0.0);
            for (int k = 0; k < j; ++k) {
                const cd lik = cplx ? cd(at(i, k)[0], at(i, k)[1]) : cd(at(i, k)[0], 0.0);
                const cd ljk = cplx ? cd(at(j, k)[0], at(j, k)[1]) : cd(at(j, k)[0], 0.0);
                v -= lik * std::conj(ljk);
            }
            v /= ljj;
            at(i, j)[0] = v.real();
            if (cplx) at(i, j)[1] = v.imag();
        }
    }
    return 0;
}

// B (n x n) <- op(L)^-1 B (side 'L') or B op(L)^-1 (side 'R'), L lower triangular n x n, op = 'N' or 'C'
void builtin_trsm_lower(bool cplx, char side, char op, int n, const double* L_, int ldl, double* B_, int ldb)
{
    const int E = cplx ? 2 : 1;
    auto L = [&](int i, int j) -> cd {
        const double* p = L_ + E * (i + (size_t)j * ldl);
        return cplx ? cd(p[0], p[1]) : cd(p[0], 0.0);
    };
    auto getB = [&](int i, int j) -> cd {
        const double* p = B_ + E * (i + (size_t)j * ldb);
        return cplx ? cd(p[0], p[1]) : cd(p[0], 0.0);
    };
    auto setB = [&](int i, int j, cd v) {
        double* p = B_ + E * (i + (size_t)j * ldb);
        p[0] = v.real();
        if (cplx) p[1] = v.imag();
    };
    const bool conj_t = (op == 'C' || op == 'c' || op == 'T' || op == 't');
    if (side == 'L' || side == 'l') {
        for (int j = 0; j < n; ++j) {
            if (!conj_t) {                             // forward substitution with L
                for (int i = 0; i < n; ++i) {
                    cd v = getB(i, j);
                    for (int k = 0; k < i; ++k) v -= L(i, k) * getB(k, j);
                    setB(i, j, v / L(i, i));
                }
            } else {                                   // backward substitution with L^H
                for (int i = n - 1; i >= 0; --i) {
                    cd v = getB(i, j);
                    for (int k = i + 1; k < n; ++k) v -= std::conj(L(k, i)) * getB(k, j);
                    setB(i, j, v / std::conj(L(i, i)));
                }
            }
        }
    } else {
        for (int i = 0; i < n; ++i) {                  // row i of B: x op(L) = b
            if (conj_t) {                              // x L^H = b  ->  forward over columns
                for (int j = 0; j < n; ++j) {
                    cd v = getB(i, j);
                    for (int k = 0; k < j; ++k) v -= getB(i, k) * std::conj(L(j, k));
                    setB(i, j, v / std::conj(L(j, j)));
                }
            } else {                                   // x L = b  ->  backward over columns
                for (int j = n - 1; j >= 0; --j) {
                    cd v = getB(i, j);
                    for (int k = j + 1; k < n; ++k) v -= getB(i, k) * L(k, j);
                    setB(i, j, v / L(j, j));
                }
            }
        }
    }
}

} // namespace chase_hip
