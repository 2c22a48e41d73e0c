// kernels.h — internal C++ declarations of the HIP kernel launchers (not part of the C ABI; see include/chase_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace chase_hip {

// C = alpha*op(A)*B + beta*C, column-major, device pointers. cplx: elements are interleaved (re,im) doubles.
// lda/ldb/ldc in elements. alpha/beta point to 1 (real) or 2 (complex) host doubles.
// ws/ws_bytes: optional device workspace for deterministic split-K.
// tag: 1 = Chebyshev-filter product (own kernel symbol; complex products may take the three-multiplication scheme),
// 2 = H-times-block product outside the filter, 0 = everything else.  device: ordinal the stream lives on (per-device
// kernel attributes).  exec_flops (optional): += the flops the matrix cores execute for this product.  min_rounds > 0: the
// launch shares the chip with a collective - products with fewer tiles than min_rounds per workgroup slot are cut along K.
int gemm_f64(hipStream_t st, bool cplx, char opA, int m, int n, int k, const double* alpha, const double* A, long lda,
             const double* B, long ldb, const double* beta, double* C, long ldc, double* ws, size_t ws_bytes,
             int num_cu, int tag = 0, int device = 0, double* exec_flops = nullptr, int min_rounds = 0);
constexpr size_t GEMM_WS_CAP = (size_t)640 << 20;         // upper bound of the split-K workspace
size_t gemm_f64_ws_need(bool cplx, char opA, int m, int n, int k, int num_cu, int min_rounds = 0);
// gemm_f64's status when the caller's split-K workspace is smaller than gemm_f64_ws_need says for the shape (the launcher
// never picks another split to fit: the summation order of a shape must not depend on allocation history)
constexpr int GEMM_F64_EWORKSPACE = -77001;
int gemm3m_enabled();
void gemm3m_set(int on);

int mfma_f64_peak(hipStream_t st, double* out, int blocks, int iters);
int stream_copy(hipStream_t st, void* dst, const void* src, size_t bytes);

// ---- streaming / reduction kernels (vec_kernels.hip); *_d arguments are in doubles (m*ept, ld*ept) ----
int shift_diag(hipStream_t st, double* H, long ld, int n, int ept, double shift);
int shift_list(hipStream_t st, double* H, long ld, const int* rows, const int* cols, int cnt, int ept, double shift);
int shift_diag_dev(hipStream_t st, double* A, long ld, int n, int ept, const double* nrmf_dev, double factor);
int abs_trace(hipStream_t st, const double* A, long ld, int n, int ept, double* out_dev);
int copy2d(hipStream_t st, const double* src, long ld_src_d, double* dst, long ld_dst_d, long md, int ncols);
int swap_cols(hipStream_t st, double* a, double* b, long md);
int rows_indexed(hipStream_t st, bool cplx, const double* in, long ld_in, double* out, long ld_out, const int* idx_dev,
                 int np, int ncols, int scatter);
int hash64(hipStream_t st, const double* x, long ld_d, long md, int ncols, unsigned long long* out_dev);
int tri_mask_bc(hipStream_t st, bool cplx, double* H, long ldh, int mloc, int nloc, long mb, int pr, int pi, long nb, int pc,
                int pj, int keep_upper);
int conj_transpose_add(hipStream_t st, bool cplx, const double* P, long ldp, int nr, int nc, const int* rowmap_dev,
                       const int* colmap_dev, double* H, long ldh);
int copy_cols_indexed(hipStream_t st, const double* src, long ld_src_d, double* dst, long ld_dst_d, long md,
                      const int* src_idx_dev, const int* dst_idx_dev, int cnt);
// dst column dst0 + c <- src column src_idx[c], c < cnt (md doubles per column)
int copy_cols_indexed_range(hipStream_t st, const double* src, long ld_src_d, double* dst, long ld_dst_d, long md,
                            const int* src_idx_dev, int dst0, int cnt);
int resid_norms(hipStream_t st, const double* W, long ldw_d, const double* V, long ldv_d, const double* lambda_dev,
                long md, int ncols, double* out_dev, int do_sqrt);
int sqrt_inplace(hipStream_t st, double* x, int n);
int col_dot(hipStream_t st, bool cplx, const double* X, long ldx_d, const double* Y, long ldy_d, int m, int ncols,
            double* out_dev);
int col_axpy(hipStream_t st, bool cplx, const double* a_dev, int a_is_real, int a_stride, double sgn, const double* X,
             long ldx_d, double* Y, long ldy_d, int m, int ncols);
int col_scal(hipStream_t st, const double* a_dev, int inv, double* X, long ldx_d, long md, int ncols);
int scale_rows(hipStream_t st, double* X, long ldx_d, long row0_d, long md, int ncols, double s);
int scale_rows_bc(hipStream_t st, double* X, long ldx_d, long m, int ncols, int ept, long g0, long nb, int p, int q, double s);
int conj_inplace(hipStream_t st, double* X, long ldx_d, int m, int ncols);
int pack_upper(hipStream_t st, const double* A, long lda, int n, int ept, double* P);
int unpack_upper(hipStream_t st, double* P, int n, int ept, double* A, long lda);
int mirror_upper(hipStream_t st, double* A, long lda, int n, int ept);
int mirror_lower(hipStream_t st, double* A, long lda, int n, int ept, int zero_diag_imag = 1);

// ---- generators (gen_kernels.hip) ----
int fill_normal(hipStream_t st, bool cplx, double* X, long ldx, int m, int n, long grow0, long gcol0, long gld,
                unsigned long long seed, int mb = 0, int pr = 1, int pi = 0);
int gen_clement(hipStream_t st, bool cplx, double* H, long ldh, int mloc, int nloc, long N, int mb, int pr, int pi,
                long roff, int nb, int pc, int pj, long coff, double scale, double perturb, unsigned long long seed);
int gen_bse(hipStream_t st, bool cplx, double* H, long ldh, int mloc, int nloc, long N, int mb, int pr, int pi, int nb,
            int pc, int pj, double dmin, double dmax, double offdiag, unsigned long long seed);

// ---- tridiagonal divide & conquer with the O(n^2) / O(n^3) parts on the device (stedc_gpu.hip); d, e on the host ----
}
struct chase_hip_ctx;
namespace chase_hip {
int stedc_gpu(chase_hip_ctx* c, int n, const double* d, const double* e, double* w_host, double* Z_dev, long ldz);

// ---- factorisation cores (factor_kernels.hip) ----
int potf2_trtri(hipStream_t st, bool cplx, double* A, long lda, int nb, int joff, double* Tinv, int* info_dev);
int trtri_diag(hipStream_t st, bool cplx, const double* R, long ldr, int n, double* Tinv);

} // namespace chase_hip
