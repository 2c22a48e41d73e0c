// capi_blas.hip — C ABI entry points for the non-GEMM kernels of the hot path and the blocked POTRF / TRSM / CholQR
// drivers built on the MFMA GEMM (include/chase_hip.h).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#include "../../include/chase_hip.h"
#include "ctx.h"
#include "kernels.h"
#include "host_lapack.h"

using namespace chase_hip;

#define HIPCHK(x)                                                                                                      \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) return hip_fail(e_, #x);                                                                 \
    } while (0)
#define KCHK(x, what)                                                                                                  \
    do {                                                                                                               \
        int e_ = (x);                                                                                                  \
        if (e_) return hip_fail((hipError_t)e_, what);                                                                 \
    } while (0)
#define RCCHK(x)                                                                                                       \
    do {                                                                                                               \
        int r_ = (x);                                                                                                  \
        if (r_) return r_;                                                                                             \
    } while (0)

namespace {
constexpr int NB = 64;

inline int ept_of(int cplx) { return cplx ? 2 : 1; }

// internal GEMM on raw double* with element-unit leading dimensions
int gemm(chase_hip_ctx* c, int cplx, char op, int m, int n, int k, double ar, double ai, const double* A, long lda,
         const double* B, long ldb, double br, double bi, double* C, long ldc)
{
    if (m <= 0 || n <= 0) return 0;
    const double alpha[2] = {ar, ai}, beta[2] = {br, bi};
    return c->gemm(cplx != 0, op, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}
} // namespace

extern "C" {

int chase_hip_set_lapack_lib(const char* path) { return lapack_bind(path); }
const char* chase_hip_lapack_provider(void)
{
    lapack_bind(nullptr);
    return lapack_provider();
}
/* binds the provider and runs tiny problems through the routines the hot path uses, so that the library's lazily loaded
 * compute kernels (MKL: seconds on a cold page cache) are resident before the first solve */
int chase_hip_host_lapack_warmup(void)
{
    int rc = lapack_bind(nullptr);
    if (rc) return rc;
    static bool done = false;
    if (done) return 0;
    const int n = 8;
    double d[n], e[n], w[n], Z[n * n];
    for (int i = 0; i < n; ++i) { d[i] = i; e[i] = 0.5; }
    rc = host_stedc(n, d, e, w, Z, n);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) { d[i] = i; e[i] = 0.5; }
    rc = host_stemr(n, d, e, w, Z, n);
    if (rc) return rc;
    double A[2 * n * n];
    for (int i = 0; i < 2 * n * n; ++i) A[i] = 0.0;
    for (int i = 0; i < n; ++i) A[2 * (i + i * n)] = 1.0 + i;
    rc = host_heevd(true, n, A, n, w);
    if (rc) return rc;
    done = true;
    return 0;
}

int chase_hip_set_host_threads(int n)
{
    lapack_bind(nullptr);
    lapack_set_threads(n);
    return 0;
}

int chase_hip_ctx_set_phase(chase_hip_ctx* c, int phase)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "set_phase: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    c->phase = phase;
    return 0;
}

int chase_hip_ctx_set_gemm_min_rounds(chase_hip_ctx* c, int rounds)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx_set_gemm_min_rounds: NULL context");
    c->gemm_min_rounds = rounds < 0 ? 0 : (rounds > 16 ? 16 : rounds);
    return 0;
}

/* X (m x n local window whose first element is global (grow0, gcol0) of a matrix with gld global rows) ~ N(0,1) */
int chase_hip_fill_normal(chase_hip_ctx* c, int cplx, int m, int n, void* X, long ldx, long grow0, long gcol0, long gld,
                          unsigned long long seed)
{
    if (!c || !X) return set_error(CHASE_HIP_EINVAL, "fill_normal: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("fill_normal", m, n, 0, 0);
    if (m < 0 || n < 0 || ldx < m) return set_error(CHASE_HIP_EINVAL, "fill_normal: bad shape");
    KCHK(fill_normal(c->stream, cplx != 0, (double*)X, ldx, m, n, grow0, gcol0, gld, seed), "fill_normal");
    return 0;
}

/* same, rows of the local window are the block-cyclic rows (mb, pr, pi) of the global matrix */
int chase_hip_fill_normal_bc(chase_hip_ctx* c, int cplx, int m, int n, void* X, long ldx, long gld, int mb, int pr,
                             int pi, unsigned long long seed)
{
    if (!c || !X) return set_error(CHASE_HIP_EINVAL, "fill_normal_bc: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("fill_normal_bc", m, n, 0, 0);
    if (m < 0 || n < 0 || ldx < m || mb <= 0 || pr <= 0) return set_error(CHASE_HIP_EINVAL, "fill_normal_bc: bad shape");
    KCHK(fill_normal(c->stream, cplx != 0, (double*)X, ldx, m, n, 0, 0, gld, seed, mb, pr, pi), "fill_normal_bc");
    return 0;
}

/* rows by index list: scatter == 0: out[p,:] = in[idx[p],:];  scatter != 0: out[idx[p],:] = in[p,:]  (idx on device) */
int chase_hip_rows_indexed(chase_hip_ctx* c, int cplx, const void* in, long ld_in, void* out, long ld_out,
                           const int* idx_dev, int np, int ncols, int scatter)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "rows_indexed: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("rows_indexed", np, ncols, scatter, 0);
    KCHK(rows_indexed(c->stream, cplx != 0, (const double*)in, ld_in, (double*)out, ld_out, idx_dev, np, ncols, scatter),
         "rows_indexed");
    return 0;
}

/* local shard (mloc x nloc) of the N x N Clement-type test matrix.  Rows: global = roff + ((l / mb) * pr + pi) * mb +
 * l % mb; columns likewise with (nb, pc, pj, coff).  Whole matrix on one GPU: mb = nb = N, pr = pc = 1, rest 0.
 * H = scale * (Clement + perturb * Hermitian N(0,1)); perturb = 0 gives the unperturbed tridiagonal matrix. */
/* 64-bit content hash of a device matrix (m x n, ld; position-mixed words summed modulo 2^64: order-independent, so
 * reproducible) - lets two holders of what should be the same block compare it without moving it */
int chase_hip_hash64(chase_hip_ctx* c, int cplx, int m, int n, const void* A, long lda, unsigned long long* out_host)
{
    if (!c || !out_host) return set_error(CHASE_HIP_EINVAL, "hash64: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (m < 0 || n < 0 || lda < m) return set_error(CHASE_HIP_EINVAL, "hash64: bad shape");
    *out_host = 0;
    if (m == 0 || n == 0) return 0;
    if (!A) return set_error(CHASE_HIP_EINVAL, "hash64: NULL matrix");
    const int e = ept_of(cplx);
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_SCAL, 4096));
    unsigned long long* d = (unsigned long long*)c->bufs[chase_hip_ctx::BUF_SCAL];
    HIPCHK(hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream));
    KCHK(hash64(c->stream, (const double*)A, lda * e, (long)m * e, n, d), "hash64");
    HIPCHK(hipMemcpyAsync(out_host, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

/* out[:, c] = in[:, idx[c]], c < ncols (idx on the device): column gather, the counterpart of chase_hip_rows_indexed */
int chase_hip_cols_indexed(chase_hip_ctx* c, int cplx, int m, const void* in, long ld_in, void* out, long ld_out,
                           const int* idx_dev, int ncols)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "cols_indexed: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("cols_indexed", m, ncols, 0, 0);
    if (m < 0 || ncols < 0 || ld_in < m || ld_out < m) return set_error(CHASE_HIP_EINVAL, "cols_indexed: bad shape");
    if (m == 0 || ncols == 0) return 0;
    if (!in || !out || !idx_dev) return set_error(CHASE_HIP_EINVAL, "cols_indexed: NULL argument");
    const int e = ept_of(cplx);
    KCHK(copy_cols_indexed_range(c->stream, (const double*)in, ld_in * e, (double*)out, ld_out * e, (long)m * e, idx_dev, 0, ncols),
         "cols_indexed");
    return 0;
}

/* A (n x n, device) <- Hermitian completion of its stored triangle: uplo 'U': A[i,j] = conj(A[j,i]) for i > j, 'L' the other
 * way round; the diagonal is left as it is - cpu::symOrHermMatrix (linalg/internal/cpu/symOrHerm.hpp:111-134) on a matrix that
 * lives in HBM */
int chase_hip_complete_hermitian(chase_hip_ctx* c, int cplx, char uplo, int n, void* A, long lda)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "complete_hermitian: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("complete_hermitian", n, uplo, 0, 0);
    if (uplo != 'U' && uplo != 'L' && uplo != 'u' && uplo != 'l') return set_error(CHASE_HIP_EINVAL, "complete_hermitian: uplo must be 'U' or 'L'");
    if (n < 0 || lda < n) return set_error(CHASE_HIP_EINVAL, "complete_hermitian: bad shape");
    if (n == 0) return 0;
    if (!A) return set_error(CHASE_HIP_EINVAL, "complete_hermitian: NULL matrix");
    if (uplo == 'U' || uplo == 'u') KCHK(mirror_upper(c->stream, (double*)A, lda, n, ept_of(cplx)), "mirror_upper");
    else KCHK(mirror_lower(c->stream, (double*)A, lda, n, ept_of(cplx), 0), "mirror_lower");
    return 0;
}

/* Triangle mask of a block-cyclic shard: entries of the triangle that is not kept (uplo 'U': the strictly lower one, 'L': the
 * strictly upper one, by GLOBAL position) become 0, diagonal entries are halved - the first step of the distributed
 * symOrHermMatrix (linalg/internal/mpi/symOrHerm.hpp:138-170,232-296) */
int chase_hip_tri_mask_bc(chase_hip_ctx* c, int cplx, char uplo, int mloc, int nloc, void* H, long ldh, long mb, int pr, int pi,
                          long nb, int pc, int pj)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "tri_mask_bc: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("tri_mask_bc", mloc, nloc, uplo, 0);
    if (uplo != 'U' && uplo != 'L' && uplo != 'u' && uplo != 'l') return set_error(CHASE_HIP_EINVAL, "tri_mask_bc: uplo must be 'U' or 'L'");
    if (mloc < 0 || nloc < 0 || ldh < mloc || mb <= 0 || nb <= 0 || pr <= 0 || pc <= 0 || pi < 0 || pi >= pr || pj < 0 || pj >= pc)
        return set_error(CHASE_HIP_EINVAL, "tri_mask_bc: bad shape / layout");
    if (mloc == 0 || nloc == 0) return 0;
    if (!H) return set_error(CHASE_HIP_EINVAL, "tri_mask_bc: NULL matrix");
    KCHK(tri_mask_bc(c->stream, cplx != 0, (double*)H, ldh, mloc, nloc, mb, pr, pi, nb, pc, pj, (uplo == 'U' || uplo == 'u') ? 1 : 0),
         "tri_mask_bc");
    return 0;
}

/* H[colmap[b], rowmap[a]] += conj(P[a, b]), a < nr, b < nc (maps on the device): adds the conjugate transpose of a packed
 * piece into a shard - replaces the p?tranc + local add of the reference's symOrHermMatrix (mpi/symOrHerm.hpp:176-186) */
int chase_hip_conj_transpose_add(chase_hip_ctx* c, int cplx, int nr, int nc, const void* P, long ldp, const int* rowmap_dev,
                                 const int* colmap_dev, void* H, long ldh)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "conj_transpose_add: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("conj_transpose_add", nr, nc, 0, 0);
    if (nr < 0 || nc < 0 || ldp < nr) return set_error(CHASE_HIP_EINVAL, "conj_transpose_add: bad shape");
    if (nr == 0 || nc == 0) return 0;
    if (!P || !H || !rowmap_dev || !colmap_dev) return set_error(CHASE_HIP_EINVAL, "conj_transpose_add: NULL argument");
    KCHK(conj_transpose_add(c->stream, cplx != 0, (const double*)P, ldp, nr, nc, rowmap_dev, colmap_dev, (double*)H, ldh),
         "conj_transpose_add");
    return 0;
}

int chase_hip_gen_clement(chase_hip_ctx* c, int cplx, void* H, long ldh, int mloc, int nloc, long N, int mb, int pr,
                          int pi, long roff, int nb, int pc, int pj, long coff, double scale, double perturb,
                          unsigned long long seed)
{
    if (!c || !H) return set_error(CHASE_HIP_EINVAL, "gen_clement: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (mloc < 0 || nloc < 0 || ldh < mloc || mb <= 0 || nb <= 0 || pr <= 0 || pc <= 0)
        return set_error(CHASE_HIP_EINVAL, "gen_clement: bad shape");
    KCHK(gen_clement(c->stream, cplx != 0, (double*)H, ldh, mloc, nloc, N, mb, pr, pi, roff, nb, pc, pj, coff, scale,
                     perturb, seed),
         "gen_clement");
    return 0;
}

int chase_hip_gen_bse(chase_hip_ctx* c, int cplx, void* H, long ldh, int mloc, int nloc, long N, int mb, int pr, int pi,
                      int nb, int pc, int pj, double dmin, double dmax, double offdiag, unsigned long long seed)
{
    if (!c || !H) return set_error(CHASE_HIP_EINVAL, "gen_bse: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (mloc < 0 || nloc < 0 || ldh < mloc || mb <= 0 || nb <= 0 || pr <= 0 || pc <= 0 || N <= 0 || N % 2)
        return set_error(CHASE_HIP_EINVAL, "gen_bse: bad shape (N must be even)");
    KCHK(gen_bse(c->stream, cplx != 0, (double*)H, ldh, mloc, nloc, N, mb, pr, pi, nb, pc, pj, dmin, dmax, offdiag, seed),
         "gen_bse");
    return 0;
}

int chase_hip_shift_diag(chase_hip_ctx* c, int cplx, int n, void* H, long ldh, double shift)
{
    if (!c || (!H && n > 0)) return set_error(CHASE_HIP_EINVAL, "shift_diag: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("shift_diag", n, 0, 0, 0);
    if (n < 0 || ldh < n) return set_error(CHASE_HIP_EINVAL, "shift_diag: bad shape");
    KCHK(shift_diag(c->stream, (double*)H, ldh, n, ept_of(cplx), shift), "shift_diag");
    return 0;
}

int chase_hip_shift_list(chase_hip_ctx* c, int cplx, void* H, long ldh, const int* rows_dev, const int* cols_dev, int cnt,
                         double shift)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "shift_list: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("shift_list", cnt, 0, 0, 0);
    if (cnt < 0) return set_error(CHASE_HIP_EINVAL, "shift_list: negative count");
    KCHK(shift_list(c->stream, (double*)H, ldh, rows_dev, cols_dev, cnt, ept_of(cplx), shift), "shift_list");
    return 0;
}

int chase_hip_lacpy(chase_hip_ctx* c, int cplx, int m, int n, const void* A, long lda, void* B, long ldb)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "lacpy: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("lacpy", m, n, 0, 0);
    if (m < 0 || n < 0 || lda < m || ldb < m) return set_error(CHASE_HIP_EINVAL, "lacpy: bad shape");
    if (m == 0 || n == 0) return 0;
    const int e = ept_of(cplx);
    KCHK(copy2d(c->stream, (const double*)A, lda * e, (double*)B, ldb * e, (long)m * e, n), "lacpy");
    return 0;
}

int chase_hip_swap_cols(chase_hip_ctx* c, int cplx, int m, void* V, long ldv, long i, long j)
{
    if (!c || !V) return set_error(CHASE_HIP_EINVAL, "swap_cols: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("swap_cols", m, 0, 0, 0);
    if (i == j) return 0;
    const int e = ept_of(cplx);
    double* v = (double*)V;
    KCHK(swap_cols(c->stream, v + i * ldv * e, v + j * ldv * e, (long)m * e), "swap_cols");
    return 0;
}

/* V[:, dst[c]] <- V[:, src[c]] for all c simultaneously (scratch: m x (max dst + 1) device matrix, e.g. the second
 * vector buffer).  src/dst are HOST int arrays.  Applies a batch of deferred column swaps in two launches. */
int chase_hip_permute_cols(chase_hip_ctx* c, int cplx, int m, void* V, long ldv, void* scratch, long lds_,
                           const int* src_host, const int* dst_host, int cnt)
{
    if (!c || !V || !scratch) return set_error(CHASE_HIP_EINVAL, "permute_cols: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("permute_cols", m, cnt, 0, 0);
    if (cnt <= 0) return 0;
    const int e = ept_of(cplx);
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_LAMBDA, (size_t)2 * cnt * sizeof(int) + 64));
    int* ds = (int*)c->bufs[chase_hip_ctx::BUF_LAMBDA];
    int* dd = ds + cnt;
    HIPCHK(hipMemcpyAsync(ds, src_host, (size_t)cnt * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dd, dst_host, (size_t)cnt * sizeof(int), hipMemcpyHostToDevice, c->stream));
    // the host index arrays may be reused by the caller right after we return
    HIPCHK(hipStreamSynchronize(c->stream));
    KCHK(copy_cols_indexed(c->stream, (const double*)V, ldv * e, (double*)scratch, lds_ * e, (long)m * e, ds, dd, cnt),
         "permute gather");
    KCHK(copy_cols_indexed(c->stream, (const double*)scratch, lds_ * e, (double*)V, ldv * e, (long)m * e, dd, dd, cnt),
         "permute scatter");
    return 0;
}

/* strided host <-> device matrix transfer (synchronous) */
int chase_hip_upload_matrix(chase_hip_ctx* c, int cplx, int m, int n, const void* host, long ldh, void* dev, long ldd)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "upload_matrix: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("upload_matrix", m, n, 0, 0);
    if (m <= 0 || n <= 0) return 0;
    const size_t es = sizeof(double) * ept_of(cplx);
    HIPCHK(hipMemcpy2DAsync(dev, (size_t)ldd * es, host, (size_t)ldh * es, (size_t)m * es, n, hipMemcpyHostToDevice,
                            c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int chase_hip_download_matrix(chase_hip_ctx* c, int cplx, int m, int n, const void* dev, long ldd, void* host, long ldh)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "download_matrix: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("download_matrix", m, n, 0, 0);
    if (m <= 0 || n <= 0) return 0;
    const size_t es = sizeof(double) * ept_of(cplx);
    HIPCHK(hipMemcpy2DAsync(host, (size_t)ldh * es, dev, (size_t)ldd * es, (size_t)m * es, n, hipMemcpyDeviceToHost,
                            c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int chase_hip_scale_rows(chase_hip_ctx* c, int cplx, int m, int n, void* X, long ldx, int row0, double s)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "scale_rows: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("scale_rows", m, n, row0, 0);
    const int e = ept_of(cplx);
    KCHK(scale_rows(c->stream, (double*)X, ldx * e, (long)row0 * e, (long)m * e, n, s), "scale_rows");
    return 0;
}

int chase_hip_scale_rows_bc(chase_hip_ctx* c, int cplx, int m, int n, void* X, long ldx, long g0, long nb, int p, int q,
                            double s)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "scale_rows_bc: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("scale_rows_bc", m, n, 0, 0);
    if (nb <= 0 || p <= 0 || q < 0 || q >= p || (n > 0 && ldx < m)) return set_error(CHASE_HIP_EINVAL, "scale_rows_bc: bad layout");
    const int e = ept_of(cplx);
    KCHK(scale_rows_bc(c->stream, (double*)X, ldx * e, (long)m, n, e, g0, nb, p, q, s), "scale_rows_bc");
    return 0;
}

int chase_hip_conj(chase_hip_ctx* c, int m, int n, void* X, long ldx)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "conj: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("conj", m, n, 0, 0);
    KCHK(conj_inplace(c->stream, (double*)X, ldx * 2, m, n), "conj");
    return 0;
}

/* C (n x n) = A^H B for k x n operands whose product is Hermitian (A = B: the Gram matrix of CholQR; A = H Q, B = Q: the
 * projected matrix of Rayleigh-Ritz) - what cublasTsyherk / herkx compute (cuda/cholqr.hpp:110-112).  Only the tiles that
 * touch the upper triangle are multiplied: the result is built block column by block column, column block J from the
 * rows 0 .. end(J) of op(A) only, i.e. (1 + 1/nblocks) / 2 of the flops of the full product (55 % at n = 2560 with blocks
 * of 256 columns; every piece is K-split over the chip like any short-and-fat product).  mirror != 0 rebuilds the strictly
 * lower triangle from the upper one; otherwise it is left untouched.  Small products (n < 4 blocks) take the one GEMM. */
int chase_hip_herkx(chase_hip_ctx* c, int cplx, int n, int k, const void* A_, long lda, const void* B_, long ldb, void* C_,
                    long ldc, int mirror)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "herkx: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("herkx", n, k, mirror, 0);
    if (n < 0 || k < 0 || lda < k || ldb < k || ldc < n) return set_error(CHASE_HIP_EINVAL, "herkx: bad shape");   // k = 0: C = 0
    if (n == 0) return 0;
    const int e = ept_of(cplx);
    const double *A = (const double*)A_, *B = (const double*)B_;
    double* C = (double*)C_;
    static const int wb_env = [] { const char* s = getenv("CHASE_HIP_HERK_BLOCK"); return s ? atoi(s) : 256; }();
    const int wb = wb_env > 0 ? (wb_env + 127) / 128 * 128 : 0;       // whole row tiles of the product (0: one GEMM)
    if (wb == 0 || n < 4 * wb) {
        RCCHK(gemm(c, cplx, 'C', n, n, k, 1.0, 0.0, A, lda, B, ldb, 0.0, 0.0, C, ldc));
    } else {
        for (int j0 = 0; j0 < n; j0 += wb) {
            const int w = (n - j0 < wb) ? n - j0 : wb, rows = j0 + w;
            RCCHK(gemm(c, cplx, 'C', rows, w, k, 1.0, 0.0, A, lda, B + (long)j0 * ldb * e, ldb, 0.0, 0.0, C + (long)j0 * ldc * e, ldc));
        }
    }
    if (mirror) KCHK(mirror_upper(c->stream, C, ldc, n, e), "mirror_upper");
    return 0;
}

/* A(n x n) = V^H V, V is k x n: upper triangle computed (what CholQR consumes), lower triangle mirrored from it */
int chase_hip_herk(chase_hip_ctx* c, int cplx, int n, int k, const void* V, long ldv, void* A, long lda)
{
    return chase_hip_herkx(c, cplx, n, k, V, ldv, V, ldv, A, lda, 1);
}

int chase_hip_abs_trace(chase_hip_ctx* c, int cplx, int n, const void* A, long lda, double* out_host)
{
    if (!c || !out_host) return set_error(CHASE_HIP_EINVAL, "abs_trace: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("abs_trace", n, 0, 0, 0);
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_SCAL, 4096));
    double* d = (double*)c->bufs[chase_hip_ctx::BUF_SCAL];
    KCHK(abs_trace(c->stream, (const double*)A, lda, n, ept_of(cplx), d), "abs_trace");
    HIPCHK(hipMemcpyAsync(out_host, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

/* In-place upper Cholesky A = R^H R of the device matrix A (n x n); strictly-lower part untouched.
 * Returns 0 or the LAPACK info (index of the first non-positive pivot, 1-based). */
int chase_hip_potrf_upper(chase_hip_ctx* c, int cplx, int n, void* A_, long lda)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "potrf: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("potrf_upper", n, 0, 0, 0);
    if (n < 0 || lda < n) return set_error(CHASE_HIP_EINVAL, "potrf: bad shape");
    if (n == 0) return 0;
    const int e = ept_of(cplx);
    double* A = (double*)A_;
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_TINV, (size_t)NB * NB * sizeof(double) * e));
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_PANEL, (size_t)NB * n * sizeof(double) * e));
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_SCAL, 4096));
    double* Tinv = (double*)c->bufs[chase_hip_ctx::BUF_TINV];
    double* P = (double*)c->bufs[chase_hip_ctx::BUF_PANEL];
    int* info_dev = (int*)((char*)c->bufs[chase_hip_ctx::BUF_SCAL] + 2048);
    HIPCHK(hipMemsetAsync(info_dev, 0, sizeof(int), c->stream));
    for (int j0 = 0; j0 < n; j0 += NB) {
        const int nb = (n - j0 < NB) ? n - j0 : NB;
        double* Ajj = A + ((long)j0 * lda + j0) * e;
        KCHK(potf2_trtri(c->stream, cplx != 0, Ajj, lda, nb, j0, Tinv, info_dev), "potf2");
        const int rest = n - j0 - nb;
        if (rest > 0) {
            double* Ajr = A + ((long)(j0 + nb) * lda + j0) * e;            // A[j0 : j0+nb, j0+nb : n]
            // row panel  R_jr = R_jj^{-H} A_jr
            RCCHK(gemm(c, cplx, 'C', nb, rest, nb, 1.0, 0.0, Tinv, NB, Ajr, lda, 0.0, 0.0, P, nb));
            KCHK(copy2d(c->stream, P, (long)nb * e, Ajr, lda * e, (long)nb * e, rest), "potrf panel copy");
            // trailing update  A_rr -= R_jr^H R_jr
            double* Arr = A + ((long)(j0 + nb) * lda + (j0 + nb)) * e;
            RCCHK(gemm(c, cplx, 'C', rest, rest, nb, -1.0, 0.0, Ajr, lda, Ajr, lda, 1.0, 0.0, Arr, lda));
        }
    }
    int info = 0;
    HIPCHK(hipMemcpyAsync(&info, info_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return info;
}

/* V (m x n) <- V R^{-1}, R upper triangular n x n (device).  Blocked: diagonal-block inverses + MFMA GEMM sweeps.
 * Round 4: LEFT-looking over column blocks - block J first receives the contribution of all finished columns in ONE product
 * with a long inner dimension, V_J -= X(:, 0:J0) R(0:J0, J), then X_J = V_J R_JJ^{-1} (blocks wider than 64: 64-column steps
 * with rank-64 updates inside the block).  Round 3 was right-looking: each of the 40 steps read and wrote ALL remaining columns
 * of V with an inner dimension of 64 (107 GB of traffic at config 4: 41.6 ms); left-looking moves 13 GB: 26.5 ms with blocks
 * of 64 (one column tile per row panel = one round of the chip), 26.8 / 27.7 / 29.5 ms with 128 / 256 / 512
 * (CHASE_HIP_TRSM_BLOCK, a multiple of 64). */
int chase_hip_trsm_right_upper(chase_hip_ctx* c, int cplx, int m, int n, const void* R_, long ldr, void* V_, long ldv)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "trsm: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("trsm_right_upper", m, n, 0, 0);
    if (m < 0 || n < 0 || ldr < n || ldv < m) return set_error(CHASE_HIP_EINVAL, "trsm: bad shape");
    if (m == 0 || n == 0) return 0;
    const int e = ept_of(cplx);
    const double* R = (const double*)R_;
    double* V = (double*)V_;
    const int nblk = (n + NB - 1) / NB;
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_TINV, (size_t)nblk * NB * NB * sizeof(double) * e));
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_PANEL, (size_t)NB * (m > n ? m : n) * sizeof(double) * e));
    double* Tinv = (double*)c->bufs[chase_hip_ctx::BUF_TINV];
    double* P = (double*)c->bufs[chase_hip_ctx::BUF_PANEL];
    KCHK(trtri_diag(c->stream, cplx != 0, R, ldr, n, Tinv), "trtri_diag");
    static const int wb_env = [] { const char* s = getenv("CHASE_HIP_TRSM_BLOCK"); return s ? atoi(s) : 64; }();
    const int WB = (wb_env < NB) ? NB : (wb_env / NB) * NB;
    for (int J0 = 0; J0 < n; J0 += WB) {
        const int wb = (n - J0 < WB) ? n - J0 : WB, Jend = J0 + wb;
        // everything the finished columns contribute to this block, in one product
        if (J0 > 0)
            RCCHK(gemm(c, cplx, 'N', m, wb, J0, -1.0, 0.0, V, ldv, R + (long)J0 * ldr * e, ldr, 1.0, 0.0, V + (long)J0 * ldv * e, ldv));
        for (int j0 = J0; j0 < Jend; j0 += NB) {
            const int b = j0 / NB;
            const int nb = (Jend - j0 < NB) ? Jend - j0 : NB;
            double* Vj = V + (long)j0 * ldv * e;
            // X_j = V_j * T_j      (V_j already carries the updates of the previous columns)
            RCCHK(gemm(c, cplx, 'N', m, nb, nb, 1.0, 0.0, Vj, ldv, Tinv + (long)b * NB * NB * e, NB, 0.0, 0.0, P, m));
            KCHK(copy2d(c->stream, P, (long)m * e, Vj, ldv * e, (long)m * e, nb), "trsm panel copy");
            const int rest = Jend - j0 - nb;                   // the block's remaining columns only
            if (rest > 0) {
                const double* Rjr = R + ((long)(j0 + nb) * ldr + j0) * e;
                double* Vr = V + (long)(j0 + nb) * ldv * e;
                RCCHK(gemm(c, cplx, 'N', m, rest, nb, -1.0, 0.0, P, m, Rjr, ldr, 1.0, 0.0, Vr, ldv));
            }
        }
    }
    return 0;
}

/* Cholesky-QR of V (m x n) in place; A is an n x n device work matrix (lda >= n) that receives the last R.
 * variant: 1 = CholQR1, 2 = CholQR2, 3 = shifted CholQR2.  m_global is the global row count used in the shift
 * sqrt(m) * sum|A_ii| * eps (== m on a single GPU).  Returns potrf info like the reference
 * (linalg/internal/cpu/cholqr1.hpp:41-189): nonzero info of the FIRST factorisation aborts before any trsm. */
int chase_hip_cholqr(chase_hip_ctx* c, int cplx, int m, int n, void* V, long ldv, void* A, long lda, int variant,
                     long m_global)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "cholqr: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("cholqr", m, n, variant, 0);
    if (variant < 1 || variant > 3) return set_error(CHASE_HIP_EINVAL, "cholqr: variant must be 1, 2 or 3");
    if (n == 0 || m == 0) return 0;
    int info;
    RCCHK(chase_hip_herk(c, cplx, n, m, V, ldv, A, lda));
    if (variant == 3) {
        double nrmf = 0.0;
        RCCHK(chase_hip_abs_trace(c, cplx, n, A, lda, &nrmf));
        const double shift = std::sqrt((double)m_global) * nrmf * std::numeric_limits<double>::epsilon();
        RCCHK(chase_hip_shift_diag(c, cplx, n, A, lda, shift));
    }
    info = chase_hip_potrf_upper(c, cplx, n, A, lda);
    if (info != 0) return info;
    RCCHK(chase_hip_trsm_right_upper(c, cplx, m, n, A, lda, V, ldv));
    const int extra = (variant == 1) ? 0 : (variant == 2 ? 1 : 2);
    for (int r = 0; r < extra; ++r) {
        RCCHK(chase_hip_herk(c, cplx, n, m, V, ldv, A, lda));
        info = chase_hip_potrf_upper(c, cplx, n, A, lda);
        if (info < 0) return info;
        RCCHK(chase_hip_trsm_right_upper(c, cplx, m, n, A, lda, V, ldv));
    }
    return info;
}

/* resid[j] = || W[:,j] - lambda[j] * V[:,j] ||_2 (or the squared sum when squared != 0), j < n.
 * lambda_host / resid_host are host arrays of length n. */
int chase_hip_resid_norms(chase_hip_ctx* c, int cplx, int m, int n, const void* W, long ldw, const void* V, long ldv,
                          const double* lambda_host, double* resid_host, int squared)
{
    if (!c || !resid_host || (V && !lambda_host)) return set_error(CHASE_HIP_EINVAL, "resid_norms: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("resid_norms", m, n, squared, 0);
    if (n <= 0) return 0;
    const int e = ept_of(cplx);
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_LAMBDA, (size_t)2 * n * sizeof(double)));
    double* dl = (double*)c->bufs[chase_hip_ctx::BUF_LAMBDA];
    double* dr = dl + n;
    if (V) HIPCHK(hipMemcpyAsync(dl, lambda_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    KCHK(resid_norms(c->stream, (const double*)W, ldw * e, (const double*)V, ldv * e, dl, (long)m * e, n, dr,
                     squared ? 0 : 1), "resid_norms");
    HIPCHK(hipMemcpyAsync(resid_host, dr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

/* the same with the result left in DEVICE memory (out_dev: n doubles): the distributed Impl sums the squares over the row
 * group on the device and reads them back once (linalg/internal/nccl/residuals.hpp:28-88 keeps them on the device as well) */
int chase_hip_resid_norms_dev(chase_hip_ctx* c, int cplx, int m, int n, const void* W, long ldw, const void* V, long ldv,
                              const double* lambda_host, double* out_dev, int squared)
{
    if (!c || !out_dev || (V && !lambda_host)) return set_error(CHASE_HIP_EINVAL, "resid_norms_dev: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("resid_norms_dev", m, n, squared, 0);
    if (n <= 0) return 0;
    const int e = ept_of(cplx);
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_LAMBDA, (size_t)2 * n * sizeof(double)));
    double* dl = (double*)c->bufs[chase_hip_ctx::BUF_LAMBDA];
    // lambda_host is pageable host memory: the copy is staged by the runtime before the call returns
    if (V) HIPCHK(hipMemcpyAsync(dl, lambda_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    KCHK(resid_norms(c->stream, (const double*)W, ldw * e, (const double*)V, ldv * e, dl, (long)m * e, n, out_dev,
                     squared ? 0 : 1), "resid_norms");
    return 0;
}

/* Hermitian eigendecomposition of the device matrix A (n x n, lower triangle referenced) on the HOST
 * (north star: "small HEEV on host"; reference lapackpp::t_heevd 'V','L').  Eigenvalues ascending to w_host,
 * eigenvectors overwrite A on the device. */
extern "C" int chase_hip_heevd_gpu(chase_hip_ctx* c, int cplx, int n, void* A, long lda, double* w_host);

int chase_hip_heevd(chase_hip_ctx* c, int cplx, int n, void* A, long lda, double* w_host)
{
    if (!c || !w_host) return set_error(CHASE_HIP_EINVAL, "heevd: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("heevd", n, 0, 0, 0);
    if (n < 0 || lda < n) return set_error(CHASE_HIP_EINVAL, "heevd: bad shape");
    if (n == 0) return 0;
    {   // large projected problems: tridiagonalise and back-transform on the GPU, only the O(n^2) tridiagonal solve on
        // the host (hetrd.hip).  CHASE_HIP_HEEVD_GPU_MIN overrides the switch-over size (0 = always host).
        static int thr = -1;
        if (thr < 0) { const char* e = getenv("CHASE_HIP_HEEVD_GPU_MIN"); thr = e ? atoi(e) : 384; }
        if (thr > 0 && n >= thr) {
            // the divide & conquer stage's launches depend on the data (deflation): the operator log lists heevd only
            ++c->oplog_mute;
            const int rc = chase_hip_heevd_gpu(c, cplx, n, A, lda, w_host);
            --c->oplog_mute;
            return rc;
        }
    }
    const int e = ept_of(cplx);
    const size_t colb = (size_t)n * sizeof(double) * e;
    RCCHK(c->ensure_hstage(colb * n));
    double* h = (double*)c->hstage;
    HIPCHK(hipMemcpy2DAsync(h, colb, A, (size_t)lda * sizeof(double) * e, colb, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    RCCHK(host_heevd(cplx != 0, n, h, n, w_host));
    HIPCHK(hipMemcpy2DAsync(A, (size_t)lda * sizeof(double) * e, h, colb, colb, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

/* The dense core of the pseudo-Hermitian Rayleigh-Ritz on the device (cpu/rayleighRitz.hpp:316-383 restated with the
 * entry points of this library; the reference's GPU path: cuda/rayleighRitz.hpp:511-600).  With A = R^H R (upper Cholesky)
 * and X = R^{-1} (one right-solve on the identity): B = -X^H M X through two MFMA GEMMs, (w, Z) = heevd(B), Ritz vectors
 * X Z, Ritz values 1 / (-w), first n/2 vectors normalised - the same steps in the same order as host_pseudo_rr (lower
 * Cholesky L = R^H, three triangular solves).  A and M are consumed; the vectors come back in M. */
static int pseudo_rr_device(chase_hip_ctx* c, int cplx, int n, double* A, double* M, double* w_host)
{
    const int e = ept_of(cplx);
    const size_t bytes = (size_t)n * n * sizeof(double) * e;
    int info = chase_hip_potrf_upper(c, cplx, n, A, n);
    if (info != 0) return info;                                   // > 0: A = Q^H S H Q is not positive definite
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_RR, bytes));
    double* X = (double*)c->bufs[chase_hip_ctx::BUF_RR];
    RCCHK(chase_hip_set_identity(c, cplx, n, X, n));
    RCCHK(chase_hip_trsm_right_upper(c, cplx, n, n, A, n, X, n));                                   // X = R^{-1}
    RCCHK(gemm(c, cplx, 'N', n, n, n, 1.0, 0.0, M, n, X, n, 0.0, 0.0, A, n));                       // A <- M X   (R is spent)
    RCCHK(gemm(c, cplx, 'C', n, n, n, -1.0, 0.0, X, n, A, n, 0.0, 0.0, M, n));                      // M <- -X^H M X
    RCCHK(chase_hip_heevd(c, cplx, n, M, n, w_host));                                               // ascending w, Z in M
    RCCHK(gemm(c, cplx, 'N', n, n, n, 1.0, 0.0, X, n, M, n, 0.0, 0.0, A, n));                       // A <- X Z = L^{-H} Z
    for (int i = 0; i < n; ++i) w_host[i] = 1.0 / (-w_host[i]);
    RCCHK(c->ensure_buf(chase_hip_ctx::BUF_SCAL, 4096 + (size_t)n * sizeof(double)));
    double* nrm = (double*)((char*)c->bufs[chase_hip_ctx::BUF_SCAL] + 4096);
    RCCHK(chase_hip_col_nrm2(c, cplx, n, n / 2, A, n, nrm));
    RCCHK(chase_hip_col_scal(c, cplx, n, n / 2, nrm, 1, A, n));
    HIPCHK(hipMemcpyAsync(M, A, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

/* Pseudo-Hermitian Rayleigh-Ritz, small dense part on the host (cpu/rayleighRitz.hpp:316-383): device A = Q^H S H Q and
 * M = Q^H S Q (both n x n, ld n) -> device M = back-transformed Ritz vectors (first n/2 columns normalised), Ritz values
 * to ritzv_host.  Returns the potrf info (> 0) if A is not positive definite. */
int chase_hip_pseudo_rr_small(chase_hip_ctx* c, int cplx, int n, void* A_dev, void* M_dev, double* ritzv_host)
{
    if (!c || !A_dev || !M_dev || !ritzv_host) return set_error(CHASE_HIP_EINVAL, "pseudo_rr_small: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("pseudo_rr_small", n, 0, 0, 0);
    if (n <= 0) return 0;
    // one line for the whole dense core: its inner sequence ends early when A does not factorise (data-dependent)
    struct Mute { chase_hip_ctx* c; Mute(chase_hip_ctx* x) : c(x) { ++c->oplog_mute; } ~Mute() { --c->oplog_mute; } } mute(c);
    const size_t bytes = (size_t)n * n * sizeof(double) * ept_of(cplx);
    // large cores (config 5: 2 (nev + nex) = 640): everything on the device like the reference's GPU path
    // (linalg/internal/cuda/rayleighRitz.hpp:511-600: cusolver potrf, cublas trsm, cusolver heevd) - see pseudo_rr_device;
    // CHASE_HIP_PSEUDO_RR_DEVICE=0 keeps round 3's split (potrf and the three trsm on the host, heevd on the device)
    static const int gpu_min = [] { const char* e = getenv("CHASE_HIP_HEEVD_GPU_MIN"); return e ? atoi(e) : 384; }();
    static const bool dev_core = [] { const char* e = getenv("CHASE_HIP_PSEUDO_RR_DEVICE"); return e ? atoi(e) != 0 : true; }();
    if (gpu_min > 0 && n >= gpu_min && dev_core) return pseudo_rr_device(c, cplx, n, (double*)A_dev, (double*)M_dev, ritzv_host);
    RCCHK(c->ensure_hstage(2 * bytes));                      // the host paths work on copies of A and M
    double* hA = (double*)c->hstage;
    double* hM = (double*)((char*)c->hstage + bytes);
    HIPCHK(hipMemcpyAsync(hA, A_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(hM, M_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (gpu_min > 0 && n >= gpu_min) {
        int rc = host_pseudo_rr_pre(cplx != 0, n, hA, hM);
        if (rc) return rc;
        std::vector<double> L(hA, hA + bytes / sizeof(double));        // the eigensolver may reuse the staging buffer
        HIPCHK(hipMemcpyAsync(M_dev, hM, bytes, hipMemcpyHostToDevice, c->stream));
        RCCHK(chase_hip_heevd(c, cplx, n, M_dev, n, ritzv_host));
        RCCHK(c->ensure_hstage(bytes));
        hM = (double*)c->hstage;
        HIPCHK(hipMemcpyAsync(hM, M_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rc = host_pseudo_rr_post(cplx != 0, n, L.data(), hM, ritzv_host);
        if (rc) return rc;
    } else {
        const int rc = host_pseudo_rr(cplx != 0, n, hA, hM, ritzv_host);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(M_dev, hM, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

/* A (n x n device, ld lda) <- identity */
int chase_hip_set_identity(chase_hip_ctx* c, int cplx, int n, void* A, long lda)
{
    if (!c || !A) return set_error(CHASE_HIP_EINVAL, "set_identity: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("set_identity", n, 0, 0, 0);
    HIPCHK(hipMemset2DAsync(A, (size_t)lda * sizeof(double) * ept_of(cplx), 0, (size_t)n * sizeof(double) * ept_of(cplx), n, c->stream));
    KCHK(shift_diag(c->stream, (double*)A, lda, n, ept_of(cplx), 1.0), "set_identity");
    return 0;
}

/* host-only twin of chase_hip_heevd (exercises the bound LAPACK provider without a GPU) */
int chase_hip_heevd_host(int cplx, int n, void* A_host, long lda, double* w_host)
{
    if (!A_host || !w_host) return set_error(CHASE_HIP_EINVAL, "heevd_host: NULL argument");
    return host_heevd(cplx != 0, n, (double*)A_host, (int)lda, w_host);
}

/* host-only helper: all eigenpairs of a symmetric tridiagonal matrix (reference lapackpp::t_stemr, cpu/lanczos.hpp:188) */
int chase_hip_stemr_host(int n, double* d, double* e, double* w, double* Z, int ldz)
{
    // LAPACK's MRRR does not terminate on every non-finite input: a Lanczos recurrence that broke down is an error, not a hang
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(d[i]) || (i + 1 < n && !std::isfinite(e[i])))
            return set_error(CHASE_HIP_EINVAL, "stemr: the tridiagonal matrix has non-finite entries (Lanczos breakdown?)");
    return host_stemr(n, d, e, w, Z, ldz);
}
int chase_hip_stedc(chase_hip_ctx* c, int n, const double* d_host, const double* e_host, double* w_host, double* Z_dev, long ldz)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "stedc: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("stedc", n, 0, 0, 0);
    return stedc_gpu(c, n, d_host, e_host, w_host, Z_dev, ldz);
}

/* ---- batched column (multi-vector) level-1 kernels with device-resident scalars ------------------------------ */
int chase_hip_col_dot(chase_hip_ctx* c, int cplx, int m, int n, const void* X, long ldx, const void* Y, long ldy,
                      double* out_dev)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "col_dot: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("col_dot", m, n, 0, 0);
    const int e = ept_of(cplx);
    KCHK(col_dot(c->stream, cplx != 0, (const double*)X, ldx * e, (const double*)Y, ldy * e, m, n, out_dev), "col_dot");
    return 0;
}
int chase_hip_col_nrm2(chase_hip_ctx* c, int cplx, int m, int n, const void* X, long ldx, double* out_dev)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "col_nrm2: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("col_nrm2", m, n, 0, 0);
    const int e = ept_of(cplx);
    KCHK(resid_norms(c->stream, (const double*)X, ldx * e, nullptr, 0, nullptr, (long)m * e, n, out_dev, 1), "col_nrm2");
    return 0;
}
/* out_dev[j] = sum_i |X[i,j]|^2 (no sqrt: partial sums for a distributed norm) */
int chase_hip_col_sumsq(chase_hip_ctx* c, int cplx, int m, int n, const void* X, long ldx, double* out_dev)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "col_sumsq: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("col_sumsq", m, n, 0, 0);
    const int e = ept_of(cplx);
    KCHK(resid_norms(c->stream, (const double*)X, ldx * e, nullptr, 0, nullptr, (long)m * e, n, out_dev, 0), "col_sumsq");
    return 0;
}
int chase_hip_sqrt_inplace(chase_hip_ctx* c, double* x_dev, int n)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "sqrt_inplace: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("sqrt_inplace", n, 0, 0, 0);
    KCHK(sqrt_inplace(c->stream, x_dev, n), "sqrt_inplace");
    return 0;
}
int chase_hip_col_axpy(chase_hip_ctx* c, int cplx, int m, int n, const double* a_dev, int a_is_real, int a_stride,
                       double sgn, const void* X, long ldx, void* Y, long ldy)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "col_axpy: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("col_axpy", m, n, a_is_real, 0);
    const int e = ept_of(cplx);
    KCHK(col_axpy(c->stream, cplx != 0, a_dev, a_is_real, a_stride, sgn, (const double*)X, ldx * e, (double*)Y, ldy * e,
                  m, n), "col_axpy");
    return 0;
}
int chase_hip_col_scal(chase_hip_ctx* c, int cplx, int m, int n, const double* a_dev, int inverse, void* X, long ldx)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "col_scal: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("col_scal", m, n, inverse, 0);
    const int e = ept_of(cplx);
    KCHK(col_scal(c->stream, a_dev, inverse, (double*)X, ldx * e, (long)m * e, n), "col_scal");
    return 0;
}

/* pack / unpack the upper triangle (column-packed) — halves the Gram all-reduce payload */
int chase_hip_pack_upper(chase_hip_ctx* c, int cplx, int n, const void* A, long lda, void* P)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "pack_upper: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("pack_upper", n, 0, 0, 0);
    KCHK(pack_upper(c->stream, (const double*)A, lda, n, ept_of(cplx), (double*)P), "pack_upper");
    return 0;
}
int chase_hip_unpack_upper(chase_hip_ctx* c, int cplx, int n, const void* P, void* A, long lda, int mirror)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "unpack_upper: NULL ctx");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (c->oplog_on) c->oplog_add("unpack_upper", n, mirror, 0, 0);
    KCHK(unpack_upper(c->stream, (double*)P, n, ept_of(cplx), (double*)A, lda), "unpack_upper");
    if (mirror) KCHK(mirror_upper(c->stream, (double*)A, lda, n, ept_of(cplx)), "mirror_upper");
    return 0;
}

} // extern "C"
