// host_lapack.cpp — dlopen-bound LAPACK (LP64, Fortran calling convention).
// Providers tried in order: $CHASE_HIP_LAPACK_LIB, an explicit hint, scipy's bundled OpenBLAS (LP64, "scipy_" prefix),
// MKL's single dynamic library.  No provider (or CHASE_HIP_LAPACK_LIB=builtin) => the library's own implementations
// (host_eig_builtin.cpp: Householder + implicit QL, Cholesky, triangular solves), reported as provider "builtin", so that a
// plain C application on a box without MKL / OpenBLAS still works (the reference links a LAPACK at build time,
// external/lapackpp/lapackpp.hpp).
#include <dlfcn.h>
#include <glob.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include "host_lapack.h"
#include "ctx.h"
#include "../../include/chase_hip.h"

namespace chase_hip {

typedef void (*dsyevd_t)(const char*, const char*, const int*, double*, const int*, double*, double*, const int*, int*,
                         const int*, int*);
typedef void (*zheevd_t)(const char*, const char*, const int*, void*, const int*, double*, void*, const int*, double*,
                         const int*, int*, const int*, int*);
typedef void (*dstemr_t)(const char*, const char*, const int*, double*, double*, const double*, const double*,
                         const int*, const int*, int*, double*, double*, const int*, const int*, int*, int*, double*,
                         const int*, int*, const int*, int*);
typedef void (*dstedc_t)(const char*, const int*, double*, double*, double*, const int*, double*, const int*, int*,
                         const int*, int*);
typedef void (*potrf_t)(const char*, const int*, void*, const int*, int*);
typedef void (*trsm_t)(const char*, const char*, const char*, const char*, const int*, const int*, const void*, const void*,
                       const int*, void*, const int*);
typedef void (*set_threads_t)(int);

static void* g_handle = nullptr;
static dsyevd_t g_dsyevd = nullptr;
static zheevd_t g_zheevd = nullptr;
static dstemr_t g_dstemr = nullptr;
static dstedc_t g_dstedc = nullptr;
static potrf_t g_dpotrf = nullptr, g_zpotrf = nullptr;
static trsm_t g_dtrsm = nullptr, g_ztrsm = nullptr;
static set_threads_t g_set_threads = nullptr;
static std::string g_provider;
static std::mutex g_mu;
static bool g_builtin = false;

int builtin_tridiag_eig(int n, double* d, double* e, double* w, double* Z, int ldz);
int builtin_heevd(bool cplx, int n, double* A, int lda, double* w);
int builtin_potrf_lower(bool cplx, int n, double* A, int lda);
void builtin_trsm_lower(bool cplx, char side, char op, int n, const double* L, int ldl, double* B, int ldb);

static void* sym_any(void* h, const char* base)
{
    const std::string b(base);
    const std::string cands[] = {"scipy_" + b + "_", b + "_", b, "scipy_" + b};
    for (const auto& c : cands) {
        void* p = dlsym(h, c.c_str());
        if (p) return p;
    }
    return nullptr;
}

static bool try_lib(const char* path)
{
    if (!path || !*path) return false;
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return false;
    dsyevd_t a = (dsyevd_t)sym_any(h, "dsyevd");
    zheevd_t b = (zheevd_t)sym_any(h, "zheevd");
    dstemr_t c = (dstemr_t)sym_any(h, "dstemr");
    if (!a || !b || !c) { dlclose(h); return false; }
    g_handle = h; g_dsyevd = a; g_zheevd = b; g_dstemr = c;
    g_dstedc = (dstedc_t)sym_any(h, "dstedc");
    g_dpotrf = (potrf_t)sym_any(h, "dpotrf"); g_zpotrf = (potrf_t)sym_any(h, "zpotrf");
    g_dtrsm = (trsm_t)sym_any(h, "dtrsm");   g_ztrsm = (trsm_t)sym_any(h, "ztrsm");
    g_set_threads = (set_threads_t)dlsym(h, "scipy_openblas_set_num_threads");
    if (!g_set_threads) g_set_threads = (set_threads_t)dlsym(h, "openblas_set_num_threads");
    if (!g_set_threads) g_set_threads = (set_threads_t)dlsym(h, "MKL_Set_Num_Threads");
    g_provider = path;
    // The projected eigenproblems are small (n <= a few thousand): OpenBLAS/MKL with one thread per core of a
    // 2 x 64-core host is 4x SLOWER than 16 threads (measured on the MI355X box: zheevd n=640 215 ms vs 55 ms,
    // n=2560 4.1 s vs 1.0 s).  Override with CHASE_HIP_HOST_THREADS.
    // MKL is private to this library -> default 16 threads.  A bundled OpenBLAS may be the very instance numpy/scipy use:
    // changing its thread count behind their back crashed scipy's trsm in testing, so it is only touched on request.
    const bool is_mkl = strstr(path, "mkl") != nullptr;
    int nt = is_mkl ? 16 : 0;
    if (const char* e = getenv("CHASE_HIP_HOST_THREADS")) nt = atoi(e);
    if (g_set_threads && nt > 0) g_set_threads(nt);
    return true;
}

static bool try_glob(const char* pattern)
{
    glob_t g;
    bool ok = false;
    if (glob(pattern, 0, nullptr, &g) == 0) {
        for (size_t i = 0; i < g.gl_pathc && !ok; ++i) ok = try_lib(g.gl_pathv[i]);
    }
    globfree(&g);
    return ok;
}

int lapack_bind(const char* hint)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_handle || g_builtin) return 0;
    const char* forced = getenv("CHASE_HIP_LAPACK_LIB");
    if ((forced && strcmp(forced, "builtin") == 0) || (hint && strcmp(hint, "builtin") == 0)) {
        g_builtin = true;
        g_provider = "builtin";
        return 0;
    }
    if (try_lib(forced)) return 0;
    if (try_lib(hint)) return 0;
    // 1) MKL's single dynamic library: a PRIVATE provider (numpy/scipy in the same process use their own OpenBLAS), so
    //    its thread count can be tuned without side effects.  LP64 interface, GNU threading layer (libgomp).
    setenv("MKL_INTERFACE_LAYER", "GNU,LP64", 0);
    setenv("MKL_THREADING_LAYER", "GNU", 0);
    if (try_lib("/opt/conda/lib/libmkl_rt.so") || try_lib("libmkl_rt.so.2") || try_lib("libmkl_rt.so")) return 0;
    // 2) an OpenBLAS that Python packages bundle (shared with numpy/scipy when they are loaded: never retuned here)
    const char* pats[] = {
        "/usr/local/lib/python3*/dist-packages/scipy.libs/libscipy_openblas-*.so",
        "/usr/lib/python3*/dist-packages/scipy.libs/libscipy_openblas-*.so",
        "/opt/conda/lib/python3*/site-packages/scipy.libs/libscipy_openblas-*.so",
        "/usr/lib/x86_64-linux-gnu/liblapack.so.3",
        "/usr/lib/x86_64-linux-gnu/libopenblas.so.0",
    };
    for (const char* p : pats)
        if (try_glob(p)) return 0;
    // 3) nothing to bind: the library's own small eigensolvers (slower, same contracts)
    g_builtin = true;
    g_provider = "builtin";
    return 0;
}

const char* lapack_provider() { return g_provider.c_str(); }

void lapack_set_threads(int n)
{
    if (g_set_threads && n > 0) g_set_threads(n);
}

int host_heevd(bool cplx, int n, double* A, int lda, double* w)
{
    if (n <= 0) return 0;
    int rc = lapack_bind(nullptr);
    if (rc) return rc;
    if (g_builtin) {
        const int bi = builtin_heevd(cplx, n, A, lda, w);
        return bi ? set_error(CHASE_HIP_ENOTCONV, "builtin heevd: QL iteration did not converge") : 0;
    }
    int info = 0, lwork = -1, lrwork = -1, liwork = -1;
    const char jobz = 'V', uplo = 'L';
    if (!cplx) {
        double wq; int iwq;
        g_dsyevd(&jobz, &uplo, &n, A, &lda, w, &wq, &lwork, &iwq, &liwork, &info);
        if (info) return set_error(CHASE_HIP_ENOTCONV, "dsyevd workspace query failed");
        lwork = (int)wq; liwork = iwq;
        std::vector<double> work((size_t)lwork);
        std::vector<int> iwork((size_t)liwork);
        g_dsyevd(&jobz, &uplo, &n, A, &lda, w, work.data(), &lwork, iwork.data(), &liwork, &info);
    } else {
        double wq[2], rwq; int iwq;
        g_zheevd(&jobz, &uplo, &n, A, &lda, w, wq, &lwork, &rwq, &lrwork, &iwq, &liwork, &info);
        if (info) return set_error(CHASE_HIP_ENOTCONV, "zheevd workspace query failed");
        lwork = (int)wq[0]; lrwork = (int)rwq; liwork = iwq;
        std::vector<double> work(2 * (size_t)lwork), rwork((size_t)lrwork);
        std::vector<int> iwork((size_t)liwork);
        g_zheevd(&jobz, &uplo, &n, A, &lda, w, work.data(), &lwork, rwork.data(), &lrwork, iwork.data(), &liwork, &info);
    }
    if (info != 0) {
        char buf[128];
        snprintf(buf, sizeof buf, "host heevd failed, info = %d", info);
        return set_error(info > 0 ? CHASE_HIP_ENOTCONV : CHASE_HIP_EINVAL, buf);
    }
    return 0;
}

// Small dense core of the pseudo-Hermitian Rayleigh-Ritz (reference: cpu::rayleighRitz_v2,
// linalg/internal/cpu/rayleighRitz.hpp:316-383):  A = Q^H S H Q (Hermitian positive definite), M = Q^H S Q.
//   A = L L^H;  M <- -L^-1 M L^-H;  heevd(M) -> (w, Z);  w <- -w;  Z <- L^-H Z;  ritz = 1 / w;  normalise Z[:, :n/2]
// On return M holds Z and w the Ritz values.  Returns potrf info (> 0) if A is not positive definite.
// The dense core of the pseudo-Hermitian Rayleigh-Ritz in two halves around the eigensolver (cpu/rayleighRitz.hpp:330-392):
//   pre :  A = L L^H,  M <- -(L^-1 M L^-H)                                   (returns the potrf info if A is not positive definite)
//   post:  Z <- L^-H Z,  w <- 1 / (-w),  first n/2 columns normalised        (Z = eigenvectors of the matrix pre left in M)
int host_pseudo_rr_pre(bool cplx, int n, double* A, double* M)
{
    if (n <= 0) return 0;
    int rc = lapack_bind(nullptr);
    if (rc) return rc;
    potrf_t potrf = cplx ? g_zpotrf : g_dpotrf;
    trsm_t trsm = cplx ? g_ztrsm : g_dtrsm;
    const bool own = g_builtin || !potrf || !trsm;          // a provider without potrf / trsm: the own ones serve
    const int E = cplx ? 2 : 1;
    int info = 0;
    const double one[2] = {1.0, 0.0};
    if (own) {
        info = builtin_potrf_lower(cplx, n, A, n);
        if (info != 0) return info;
        builtin_trsm_lower(cplx, 'L', 'N', n, A, n, M, n);
        builtin_trsm_lower(cplx, 'R', 'C', n, A, n, M, n);
    } else {
        potrf("L", &n, A, &n, &info);
        if (info != 0) return info > 0 ? info : set_error(CHASE_HIP_EINVAL, "host potrf: illegal argument");
        trsm("L", "L", "N", "N", &n, &n, one, A, &n, M, &n);
        trsm("R", "L", "C", "N", &n, &n, one, A, &n, M, &n);
    }
    for (size_t i = 0; i < (size_t)n * n * E; ++i) M[i] = -M[i];
    return 0;
}
int host_pseudo_rr_post(bool cplx, int n, const double* A, double* M, double* w)
{
    if (n <= 0) return 0;
    trsm_t trsm = cplx ? g_ztrsm : g_dtrsm;
    potrf_t potrf = cplx ? g_zpotrf : g_dpotrf;
    const bool own = g_builtin || !potrf || !trsm;
    const int E = cplx ? 2 : 1;
    const double one[2] = {1.0, 0.0};
    for (int i = 0; i < n; ++i) w[i] = -w[i];
    if (own) builtin_trsm_lower(cplx, 'L', 'C', n, A, n, M, n);
    else trsm("L", "L", "C", "N", &n, &n, one, A, &n, M, &n);
    for (int i = 0; i < n; ++i) w[i] = 1.0 / w[i];
    for (int j = 0; j < n / 2; ++j) {
        double s = 0.0;
        double* col = M + (size_t)j * n * E;
        for (int i = 0; i < n * E; ++i) s += col[i] * col[i];
        const double inv = 1.0 / std::sqrt(s);
        for (int i = 0; i < n * E; ++i) col[i] *= inv;
    }
    return 0;
}
int host_pseudo_rr(bool cplx, int n, double* A, double* M, double* w)
{
    int rc = host_pseudo_rr_pre(cplx, n, A, M);
    if (rc) return rc;
    rc = host_heevd(cplx, n, M, n, w);
    if (rc) return rc;
    return host_pseudo_rr_post(cplx, n, A, M, w);
}

// divide & conquer tridiagonal eigensolver (compz = 'I'): eigenvalues ascending in d (copied to w), eigenvectors in Z
int host_stedc(int n, double* d, double* e, double* w, double* Z, int ldz)
{
    if (n <= 0) return 0;
    int rc = lapack_bind(nullptr);
    if (rc) return rc;
    if (g_builtin || !g_dstedc) return host_stemr(n, d, e, w, Z, ldz);
    const char compz = 'I';
    int info = 0, lwork = -1, liwork = -1, iwq;
    double wq;
    g_dstedc(&compz, &n, d, e, Z, &ldz, &wq, &lwork, &iwq, &liwork, &info);
    if (info) return set_error(CHASE_HIP_ENOTCONV, "dstedc workspace query failed");
    lwork = (int)wq; liwork = iwq;
    std::vector<double> work((size_t)lwork);
    std::vector<int> iwork((size_t)liwork);
    g_dstedc(&compz, &n, d, e, Z, &ldz, work.data(), &lwork, iwork.data(), &liwork, &info);
    if (info != 0) {
        char buf[128];
        snprintf(buf, sizeof buf, "host stedc failed, info = %d", info);
        return set_error(CHASE_HIP_ENOTCONV, buf);
    }
    for (int i = 0; i < n; ++i) w[i] = d[i];
    return 0;
}

int host_stemr(int n, double* d, double* e, double* w, double* Z, int ldz)
{
    if (n <= 0) return 0;
    int rc = lapack_bind(nullptr);
    if (rc) return rc;
    if (g_builtin) {
        const int bi = builtin_tridiag_eig(n, d, e, w, Z, ldz);
        return bi ? set_error(CHASE_HIP_ENOTCONV, "builtin tridiagonal QL iteration did not converge") : 0;
    }
    const char jobz = 'V', range = 'A';
    const double vl = 0, vu = 0;
    const int il = 0, iu = 0;
    int m = 0, nzc = n, tryrac = 0, info = 0, lwork = -1, liwork = -1;
    std::vector<int> isuppz(2 * (size_t)n);
    double wq; int iwq;
    g_dstemr(&jobz, &range, &n, d, e, &vl, &vu, &il, &iu, &m, w, Z, &ldz, &nzc, isuppz.data(), &tryrac, &wq, &lwork,
             &iwq, &liwork, &info);
    if (info) return set_error(CHASE_HIP_ENOTCONV, "dstemr workspace query failed");
    lwork = (int)wq; liwork = iwq;
    std::vector<double> work((size_t)lwork);
    std::vector<int> iwork((size_t)liwork);
    g_dstemr(&jobz, &range, &n, d, e, &vl, &vu, &il, &iu, &m, w, Z, &ldz, &nzc, isuppz.data(), &tryrac, work.data(),
             &lwork, iwork.data(), &liwork, &info);
    if (info != 0 || m != n) {
        char buf[128];
        snprintf(buf, sizeof buf, "host stemr failed, info = %d, m = %d of %d", info, m, n);
        return set_error(CHASE_HIP_ENOTCONV, buf);
    }
    return 0;
}

} // namespace chase_hip
