/* chase_hip.h — C ABI of the MI355X-native ChASE hot-path backend (libchase_hip.so)
 *
 * Boundary: the reference solver talks to a backend ONLY through the virtuals of
 * chase::ChaseBase<T> (/root/reference algorithm/interface.hpp:46-434).  This header is the thin C ABI the
 * C++ Impl class (chase_amd/host/chase_hip_impl.hpp) calls, and what a ChASE maintainer would bind from an
 * Impl/chase_hip plugin (INTEGRATION.md shows the stub).  Plain pointers and sizes only, no C++/torch types.
 *
 * Conventions
 *   - every entry point returns int: 0 = ok, >0 = LAPACK-style info (potrf), <0 = runtime error
 *     (CHASE_HIP_E* or -(hipError_t)); no exceptions cross the ABI.  chase_hip_last_error() gives text.
 *   - matrices are column-major; complex = interleaved (re,im) doubles ("z" entry points take void* / double[2]).
 *   - "dev" pointers are HIP device pointers; "host" pointers are ordinary host memory.
 *   - all device work is enqueued on the context's HIP stream; entry points that return scalars to the host
 *     synchronise that stream themselves, the others are asynchronous.
 */
#ifndef CHASE_HIP_H
#define CHASE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHASE_HIP_OK 0
#define CHASE_HIP_EINVAL (-1001)   /* bad argument (shape, pointer, op) */
#define CHASE_HIP_ENODEV (-1002)   /* no usable HIP device / wrong architecture */
#define CHASE_HIP_ENOMEM (-1003)
#define CHASE_HIP_ENOTCONV (-1004) /* host eigensolver / iteration failed to converge */
#define CHASE_HIP_ECOMM (-1005)    /* RCCL / transport error */
#define CHASE_HIP_ELAPACK (-1006)  /* host LAPACK provider missing */
#define CHASE_HIP_EIO (-1007)      /* matrix file missing, too small or unreadable */

typedef struct chase_hip_ctx chase_hip_ctx;

/* ---- context -------------------------------------------------------------------------------------------------- */
/* device: HIP ordinal.  stream: an existing hipStream_t to enqueue on (e.g. torch's current stream) or NULL to let the
 * context create its own non-blocking stream. */
int chase_hip_ctx_create(chase_hip_ctx** out, int device, void* stream);
int chase_hip_ctx_destroy(chase_hip_ctx* ctx);
int chase_hip_ctx_sync(chase_hip_ctx* ctx);
void* chase_hip_ctx_stream(chase_hip_ctx* ctx);
/* Operator log: with on != 0 every C-ABI operator this context executes afterwards is listed as one line "name a b c d" (its
 * name and shapes; no pointers, no scalars), '\n'-separated in chase_hip_ctx_oplog_text; on == 0 stops listing.  The launches
 * inside the projected eigensolver depend on the data (deflation) and are not listed, only "heevd n".  What the single-rank
 * replay of a multi-GPU solve is compared with, launch for launch (bench.py --replay-rank, tests/test_gpu_replay.py). */
int chase_hip_ctx_oplog(chase_hip_ctx* ctx, int on);
int chase_hip_ctx_oplog_mute(chase_hip_ctx* ctx, int delta); /* > 0: stop listing (nestable), < 0: resume */
const char* chase_hip_ctx_oplog_text(chase_hip_ctx* ctx);
const char* chase_hip_last_error(void);
/* name may be NULL.  clock_khz is the reported max engine clock. */
int chase_hip_device_info(chase_hip_ctx* ctx, int* num_cu, int* clock_khz, size_t* hbm_bytes, char* name,
                          int name_len);
/* PCI bus id ("0000:05:00.0") of the context's device: which physical GPU a rank ended up on, whatever the visibility lists */
int chase_hip_device_bus_id(chase_hip_ctx* ctx, char* out, int len);
const char* chase_hip_version(void);
/* HIP devices visible to this process (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES applied); 0 without a GPU */
int chase_hip_device_count(void);

/* ---- device memory plumbing ----------------------------------------------------------------------------------- */
int chase_hip_malloc(chase_hip_ctx* ctx, void** dev, size_t bytes);
int chase_hip_free(chase_hip_ctx* ctx, void* dev);
int chase_hip_memcpy_h2d(chase_hip_ctx* ctx, void* dev, const void* host, size_t bytes); /* synchronous */
int chase_hip_memcpy_d2h(chase_hip_ctx* ctx, void* host, const void* dev, size_t bytes); /* synchronous */
int chase_hip_memcpy_d2d(chase_hip_ctx* ctx, void* dst, const void* src, size_t bytes);  /* stream-ordered */
int chase_hip_memset(chase_hip_ctx* ctx, void* dev, int value, size_t bytes);            /* stream-ordered */
/* stream-ordered timing helpers: elapsed milliseconds of the work enqueued by fn-free bracketing */
int chase_hip_timer_start(chase_hip_ctx* ctx);
int chase_hip_timer_stop(chase_hip_ctx* ctx, float* ms); /* synchronises */

/* ---- level-3 kernels (device pointers) -------------------------------------------------------------------------
 * C = alpha*op(A)*B + beta*C, opA in {'N','C'} ('T' == 'C' for real).  Replaces the reference's
 * blaspp::t_gemm / cublasTgemm call sites: Impl/chase_cpu/chase_cpu.hpp:497-504, Impl/chase_gpu/chase_gpu.hpp:668-675,
 * linalg/internal/mpi/hemm.hpp:155-167,209-221, linalg/internal/cpu/rayleighRitz.hpp:84-88,110. */
int chase_hip_gemm_d(chase_hip_ctx* ctx, char opA, int m, int n, int k, double alpha, const double* A, long lda,
                     const double* B, long ldb, double beta, double* C, long ldc);
int chase_hip_gemm_z(chase_hip_ctx* ctx, char opA, int m, int n, int k, const double alpha[2], const void* A,
                     long lda, const void* B, long ldb, const double beta[2], void* C, long ldc);

/* bytes of workspace a product of this shape uses on a device with num_cu compute units (context-owned, grown on
 * demand; min_rounds as in chase_hip_ctx_set_gemm_min_rounds): the split-K slabs and - complex products while the
 * three-multiplication scheme is enabled - the plane of V-side operand sums the 3M kernels read (8 bytes per element of the
 * k x n operand, rounded up to 64-column tiles).  A pure function of the shape (and of the 3M switch), so that the
 * decomposition and with it the summation order never depend on allocation history.  Host-only: callable without a GPU. */
size_t chase_hip_gemm_workspace_bytes(int cplx, char opA, int m, int n, int k, int num_cu, int min_rounds);
/* 1 when complex products issued in phase 1 (chase_hip_ctx_set_phase: the Chebyshev filter) and phase 2 (the H-times-block
 * products of Rayleigh-Ritz / residuals; CHASE_HIP_GEMM3M_RR=0 keeps those on four) use the three-multiplication scheme
 * (default; CHASE_HIP_GEMM3M=0 or chase_hip_set_gemm3m(0) selects the four-multiplication kernel, the arithmetic of the
 * reference's zgemm).  It applies to launches with m a multiple of 128, k a multiple of 8 and 16-byte addressable
 * operands; other shapes, and every other product (Gram matrices, back-transforms, QR, phase 3), take the
 * four-multiplication kernel. */
int chase_hip_gemm3m_enabled(void);
int chase_hip_set_gemm3m(int on); /* process-wide run-time switch */
/* GEMM books of a context, per phase (0 other, 1 filter, 2 H-times-block outside the filter, 3 verification): flops in the reference's
 * model (2*F*m*n*k, F = 4 complex: algorithm/performance.hpp:152,250), flops the matrix cores actually executed (3/4 of
 * the model for three-multiplication launches) and the number of products.  Any output pointer may be NULL. */
int chase_hip_ctx_gemm_counters(chase_hip_ctx* ctx, int phase, double* flops_model, double* flops_executed,
                                unsigned long long* calls, int reset);
/* register-resident v_mfma_f64_16x16x4_f64 issue-rate probe: returns achieved TFLOP/s */
int chase_hip_mfma_f64_peak(chase_hip_ctx* ctx, double* tflops);
/* streaming-copy probe: achieved HBM GB/s for a bytes-sized device-to-device float4 copy */
int chase_hip_hbm_copy_peak(chase_hip_ctx* ctx, size_t bytes, double* gbps);

/* phase 1 = inside FilterPhaseStart/End: GEMMs are launched through the filter-tagged kernel symbol so that rocprofv3
 * reports the Chebyshev-filter HEMM separately; phase 2 = an H-times-block product outside the filter (Rayleigh-Ritz,
 * residuals): ordinary symbol, three multiplications like the filter (CHASE_HIP_GEMM3M_RR=0: four); phase 3 = a
 * verification product (independent residuals, the re-check of residuals that sit on the tolerance): always four
 * multiplications, the arithmetic of the reference's zgemm; 0 = everything else (always four multiplications) */
int chase_hip_ctx_set_phase(chase_hip_ctx* ctx, int phase);
/* Granularity of the following products: rounds > 0 says they share the chip with a collective on another stream (the panel
 * products of the pipelined distributed HEMM) - a product with fewer than `rounds` output tiles per workgroup slot is then cut
 * along K into that many pieces per slot, so that the CUs the collective's kernel takes displace a fraction of a tile and
 * not a whole one; 0 (default) = the automatic decomposition.  Results stay deterministic (fixed-order slab reduction). */
int chase_hip_ctx_set_gemm_min_rounds(chase_hip_ctx* ctx, int rounds);

/* ---- on-device input generators (global-index addressed, shard-safe) ------------------------------------------ */
/* N(0,1) fill (Philox4x32-10 + Box-Muller).  Replaces cuda/random_normal_distribution.cu:21-95 (initVecs on GPU) */
int chase_hip_fill_normal(chase_hip_ctx* ctx, int cplx, int m, int n, void* X, long ldx, long grow0, long gcol0,
                          long gld, unsigned long long seed);
int chase_hip_fill_normal_bc(chase_hip_ctx* ctx, int cplx, int m, int n, void* X, long ldx, long gld, int mb, int pr,
                             int pi, unsigned long long seed);
/* row gather / scatter by a device index list (column-type <-> row-type multivector redistribution,
 * linalg/distMatrix/distMultiVector.hpp:2444-2720) */
int chase_hip_rows_indexed(chase_hip_ctx* ctx, int cplx, const void* in, long ld_in, void* out, long ld_out,
                           const int* idx_dev, int np, int ncols, int scatter);
/* 64-bit content hash of a device matrix (position-mixed 8-byte words summed modulo 2^64: independent of the order of
 * evaluation, hence reproducible): two holders of what should be the same block compare 8 bytes instead of the block */
int chase_hip_hash64(chase_hip_ctx* ctx, int cplx, int m, int n, const void* A, long lda, unsigned long long* out_host);
/* column gather by a device index list: out[:, c] = in[:, idx[c]], c < ncols */
int chase_hip_cols_indexed(chase_hip_ctx* ctx, int cplx, int m, const void* in, long ld_in, void* out, long ld_out,
                           const int* idx_dev, int ncols);
/* Hermitian completion of a device matrix from its stored triangle ('U': lower <- conj(upper)^T, 'L': the other way round; the
 * diagonal is left alone) - cpu::symOrHermMatrix (linalg/internal/cpu/symOrHerm.hpp:111-134) for a matrix that lives in HBM */
int chase_hip_complete_hermitian(chase_hip_ctx* ctx, int cplx, char uplo, int n, void* A, long lda);
/* the two local steps of the distributed symOrHermMatrix (linalg/internal/mpi/symOrHerm.hpp:127-320): the triangle mask of a
 * block-cyclic shard by global position (kept triangle untouched, the other one zeroed, diagonal halved), and
 * H[colmap[b], rowmap[a]] += conj(P[a, b]) - the conjugate transpose of a received piece added into the shard */
int chase_hip_tri_mask_bc(chase_hip_ctx* ctx, int cplx, char uplo, int mloc, int nloc, void* H, long ldh, long mb, int pr, int pi,
                          long nb, int pc, int pj);
int chase_hip_conj_transpose_add(chase_hip_ctx* ctx, int cplx, int nr, int nc, const void* P, long ldp, const int* rowmap_dev,
                                 const int* colmap_dev, void* H, long ldh);
/* Clement-type test matrix of the reference's solve tests (tests/chase_serial_solve.cpp:52-90), any 2D shard:
 * H = scale * (Clement + perturb * G), G dense Hermitian N(0,1) on the entries the reference perturbs */
int chase_hip_gen_clement(chase_hip_ctx* ctx, int cplx, void* H, long ldh, int mloc, int nloc, long N, int mb, int pr,
                          int pi, long roff, int nb, int pc, int pj, long coff, double scale, double perturb,
                          unsigned long long seed);
/* Synthetic Bethe-Salpeter test matrix H = [[A, B], [-conj(B), -conj(A)]] (A Hermitian with diagonal dmin..dmax spaced uniformly in
 * the square, B symmetric, off-diagonal entries offdiag * N(0,1)), any 2D block-cyclic shard; stands in for the matrix file of
 * examples/5_bse_benchmark/5_bse_benchmark.cpp (BASELINE config 5) */
int chase_hip_gen_bse(chase_hip_ctx* ctx, int cplx, void* H, long ldh, int mloc, int nloc, long N, int mb, int pr, int pi,
                      int nb, int pc, int pj, double dmin, double dmax, double offdiag, unsigned long long seed);
/* Raw column-major binary matrix files (N x N elements of T, no header; the reference's input format):
 * chase_hip_load_matrix_shard reads this rank's (mb x nb block-cyclic; block layout: mb = nb = block length) shard of
 * the file straight into a device block — replaces Matrix::readFromBinaryFile (linalg/matrix/matrix.hpp:313-360) and the
 * BlockBlock / BlockCyclic readFromBinaryFile views (linalg/distMatrix/distMatrix.hpp:2425-2520, 3210-3330).  A file
 * smaller than N*N elements is an error, a larger one is accepted, like the reference.
 * chase_hip_save_matrix writes a device matrix (m x n, ld) as such a file (Matrix::saveToBinaryFile). */
int chase_hip_load_matrix_shard(chase_hip_ctx* ctx, const char* path, int cplx, long N, int mloc, int nloc, int mb,
                                int pr, int pi, int nb, int pc, int pj, void* dev, long ldd);
int chase_hip_save_matrix(chase_hip_ctx* ctx, const char* path, int cplx, int m, int n, const void* dev, long ldd);
/* writes this rank's block-cyclic shard into its byte ranges of the shared N x N file (created if missing, never
 * truncated: all ranks of a grid may write the same path concurrently) — distMatrix.hpp:2241-2300,3117-3200 */
int chase_hip_save_matrix_shard(chase_hip_ctx* ctx, const char* path, int cplx, long N, int mloc, int nloc, int mb,
                                int pr, int pi, int nb, int pc, int pj, const void* dev, long ldd);

/* ---- host LAPACK provider (HEEVD / STEMR stay on the host per the north star) --------------------------------- */
int chase_hip_set_lapack_lib(const char* path);   /* optional explicit LP64 LAPACK shared library */
const char* chase_hip_lapack_provider(void);      /* path of the bound provider ("" if none) */
int chase_hip_set_host_threads(int n);
/* bind the provider and page its compute kernels in (called by the solver constructors; first use of a cold MKL costs
 * seconds, which would otherwise land in the first solve) */
int chase_hip_host_lapack_warmup(void);

/* ---- O(N*n) kernels; cplx = 0 (fp64) or 1 (complex fp64); all matrix pointers are device pointers -------------- */
/* H[i,i] += shift (real part).  Replaces chase_cpu.hpp:384-389 / cuda/shiftDiagonal.cu:23-50 */
int chase_hip_shift_diag(chase_hip_ctx* ctx, int cplx, int n, void* H, long ldh, double shift);
/* H[rows[i], cols[i]] += shift for precomputed local diagonal positions (device int arrays).
 * Replaces cuda/shiftDiagonal.cu:100-149 + Impl/pchase_gpu/pchase_gpu.hpp:340-409 */
int chase_hip_shift_list(chase_hip_ctx* ctx, int cplx, void* H, long ldh, const int* rows_dev, const int* cols_dev,
                         int cnt, double shift);
/* B = A (lacpy 'A').  Replaces cuda/lacpy.cu:503-835 */
int chase_hip_lacpy(chase_hip_ctx* ctx, int cplx, int m, int n, const void* A, long lda, void* B, long ldb);
/* swap columns i and j of V.  Replaces chase_gpu.hpp:1003-1005 (cublasTswap) */
int chase_hip_swap_cols(chase_hip_ctx* ctx, int cplx, int m, void* V, long ldv, long i, long j);
/* apply a batch of deferred Swap()s: V[:, dst[c]] <- V[:, src[c]] simultaneously; src/dst are host arrays, scratch is a
 * device matrix with at least max(dst)+1 columns (the Impl passes its second vector buffer) */
int chase_hip_permute_cols(chase_hip_ctx* ctx, int cplx, int m, void* V, long ldv, void* scratch, long lds,
                           const int* src_host, const int* dst_host, int cnt);
/* strided host <-> device matrix transfers (synchronous).  Replace Hmat_->H2D() / cublasGetMatrix
 * (chase_gpu.hpp:536,1010-1018) */
int chase_hip_upload_matrix(chase_hip_ctx* ctx, int cplx, int m, int n, const void* host, long ldh, void* dev, long ldd);
int chase_hip_download_matrix(chase_hip_ctx* ctx, int cplx, int m, int n, const void* dev, long ldd, void* host,
                              long ldh);
/* X[row0:, :] *= s.  Replaces cuda/flipSign.cu:19-260 (s = -1) and scaleLowerBlockRows */
int chase_hip_scale_rows(chase_hip_ctx* ctx, int cplx, int m, int n, void* X, long ldx, int row0, double s);
/* the same for a 1D block-cyclic row distribution (block nb over p ranks, this rank q; a block layout is nb = block
 * length): local rows whose global index is >= g0 are scaled.  Replaces the distributed flipLowerHalfMatrixSign
 * (linalg/internal/mpi/flipSign.hpp, cuda_aware_mpi/flipSign.hpp) and the l_half() damping of pchase_cpu.hpp:283-300 */
int chase_hip_scale_rows_bc(chase_hip_ctx* ctx, int cplx, int m, int n, void* X, long ldx, long g0, long nb, int p, int q,
                            double s);
/* in-place conjugate (complex only).  Replaces cuda/conjugate.cu:21-70 */
int chase_hip_conj(chase_hip_ctx* ctx, int m, int n, void* X, long ldx);
/* resid[j] = ||W_j - lambda_j V_j||_2 (squared != 0: sum of squares, for the distributed all-reduce).
 * V == NULL gives plain column norms.  Replaces cuda/residuals.cu:113-296, cpu/residuals.hpp:72-79 */
int chase_hip_resid_norms(chase_hip_ctx* ctx, int cplx, int m, int n, const void* W, long ldw, const void* V, long ldv,
                          const double* lambda_host, double* resid_host, int squared);

/* the same, result left in device memory (out_dev: n doubles) - no host round trip before the distributed all-reduce
 * (linalg/internal/nccl/residuals.hpp:28-88) */
int chase_hip_resid_norms_dev(chase_hip_ctx* ctx, int cplx, int m, int n, const void* W, long ldw, const void* V, long ldv,
                              const double* lambda_host, double* out_dev, int squared);

/* ---- Cholesky-QR building blocks (replace cublasTsyherk / cusolverDnTpotrf / cublasTtrsm, cuda/cholqr.hpp:110-132) */
int chase_hip_herk(chase_hip_ctx* ctx, int cplx, int n, int k, const void* V, long ldv, void* A, long lda);
/* C (n x n) = A^H B for k x n operands whose product is Hermitian (Gram matrix: A = B; projected matrix of Rayleigh-Ritz:
 * A = H Q, B = Q; linalg/internal/cuda/cholqr.hpp:110-112 cublasTsyherk, nccl/rayleighRitz.hpp:118-129): only the block
 * columns' parts on and above the diagonal are multiplied; mirror != 0 also fills the strictly lower triangle */
int chase_hip_herkx(chase_hip_ctx* ctx, int cplx, int n, int k, const void* A, long lda, const void* B, long ldb, void* C,
                    long ldc, int mirror);
int chase_hip_abs_trace(chase_hip_ctx* ctx, int cplx, int n, const void* A, long lda, double* out_host);
/* returns 0 or LAPACK info > 0 (first non-positive pivot) */
int chase_hip_potrf_upper(chase_hip_ctx* ctx, int cplx, int n, void* A, long lda);
int chase_hip_trsm_right_upper(chase_hip_ctx* ctx, int cplx, int m, int n, const void* R, long ldr, void* V, long ldv);
/* variant 1 = cholQR1, 2 = cholQR2, 3 = shiftedcholQR2 (linalg/internal/cpu/cholqr1.hpp:41-189); returns info */
int chase_hip_cholqr(chase_hip_ctx* ctx, int cplx, int m, int n, void* V, long ldv, void* A, long lda, int variant,
                     long m_global);
/* Householder QR fallback: V <- first n columns of Q (cpu/cholqr1.hpp:203-210: geqrf + ungqr) */
int chase_hip_houseqr(chase_hip_ctx* ctx, int cplx, int m, int n, void* V, long ldv);

/* ---- Rayleigh-Ritz: host HEEVD of a device matrix ('V','L'), eigenvectors back on the device -------------------- */
int chase_hip_heevd(chase_hip_ctx* ctx, int cplx, int n, void* A, long lda, double* w_host);
/* same contract; Householder tridiagonalisation + back-transformation on the GPU, tridiagonal stemr on the host
 * (chase_hip_heevd switches to it for n >= 384; env CHASE_HIP_HEEVD_GPU_MIN) */
int chase_hip_heevd_gpu(chase_hip_ctx* ctx, int cplx, int n, void* A, long lda, double* w_host);
/* pseudo-Hermitian (BSE) Rayleigh-Ritz: small dense part of cpu::rayleighRitz_v2 (cpu/rayleighRitz.hpp:316-383) on the host */
int chase_hip_pseudo_rr_small(chase_hip_ctx* ctx, int cplx, int n, void* A_dev, void* M_dev, double* ritzv_host);
int chase_hip_set_identity(chase_hip_ctx* ctx, int cplx, int n, void* A, long lda);
int chase_hip_heevd_host(int cplx, int n, void* A_host, long lda, double* w_host); /* host-only twin (provider check) */
/* host-only: all eigenpairs of a symmetric tridiagonal (Lanczos; lapackpp::t_stemr, cpu/lanczos.hpp:188); non-finite d / e
 * (a Lanczos recurrence that broke down) is CHASE_HIP_EINVAL: LAPACK's MRRR need not terminate on such input */
int chase_hip_stemr_host(int n, double* d, double* e, double* w, double* Z, int ldz);
/* Real symmetric tridiagonal eigenproblem by divide & conquer with the O(n^2) / O(n^3) parts on the device (secular equation,
 * Gu-Eisenstat vectors, merge GEMMs; deflation and the leaves on the host): d (n), e (n-1) on the host, eigenvalues ascending
 * to w_host, eigenvectors (n x n real, ldz) to DEVICE memory.  What chase_hip_heevd_gpu uses between its tridiagonalisation and
 * back-transformation for n >= CHASE_HIP_STEDC_GPU_MIN (512).  Replaces the tridiagonal stage of cusolverDnXheevd
 * (linalg/internal/nccl/rayleighRitz.hpp:170-173). */
int chase_hip_stedc(chase_hip_ctx* ctx, int n, const double* d_host, const double* e_host, double* w_host, double* Z_dev,
                    long ldz);

/* ---- batched multi-vector level-1 kernels with device-resident scalars (Lanczos; cuda/lanczos_kernels.cu) ------- */
/* out_dev[j] = X_j^H Y_j (complex: 2 doubles per column) */
int chase_hip_col_dot(chase_hip_ctx* ctx, int cplx, int m, int n, const void* X, long ldx, const void* Y, long ldy,
                      double* out_dev);
int chase_hip_col_nrm2(chase_hip_ctx* ctx, int cplx, int m, int n, const void* X, long ldx, double* out_dev);
/* partial sums of squares + elementwise sqrt (distributed norms: cuda/lanczos_kernels.cu:433-441 batched_sqrt) */
int chase_hip_col_sumsq(chase_hip_ctx* ctx, int cplx, int m, int n, const void* X, long ldx, double* out_dev);
int chase_hip_sqrt_inplace(chase_hip_ctx* ctx, double* x_dev, int n);
/* Y_j += sgn * a[j*a_stride] * X_j; a is a device array of reals (a_is_real) or of T */
int chase_hip_col_axpy(chase_hip_ctx* ctx, int cplx, int m, int n, const double* a_dev, int a_is_real, int a_stride,
                       double sgn, const void* X, long ldx, void* Y, long ldy);
/* X_j *= a[j] (or 1/a[j] when inverse != 0), a real device array */
int chase_hip_col_scal(chase_hip_ctx* ctx, int cplx, int m, int n, const double* a_dev, int inverse, void* X, long ldx);

/* ---- packed upper triangle (Gram all-reduce payload; cuda/lacpy.cu:837-1094) ------------------------------------ */
int chase_hip_pack_upper(chase_hip_ctx* ctx, int cplx, int n, const void* A, long lda, void* P);
int chase_hip_unpack_upper(chase_hip_ctx* ctx, int cplx, int n, const void* P, void* A, long lda, int mirror);

#ifdef __cplusplus
}
#endif
#endif /* CHASE_HIP_H */
