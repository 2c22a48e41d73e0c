// vec_kernels.hip — HBM-bound O(N*n) kernels of the ChASE hot path (wave64 shuffle reductions, 16-byte accesses)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include "kernels.h"
#include "ctx.h"
#include "../../include/chase_hip.h"

int chase_hip_ctx::gemm(bool cplx, char opA, int m, int n, int k, const double* alpha, const double* A, long lda,
                        const double* B, long ldb, const double* beta, double* C, long ldc)
{
    if (m <= 0 || n <= 0) return 0;
    if (oplog_on) oplog_add(cplx ? (opA == 'N' ? "gemm_zN" : "gemm_zC") : (opA == 'N' ? "gemm_dN" : "gemm_dC"), m, n, k,
                            phase * 100 + gemm_min_rounds);
    // workspace: what this shape's tail split can use (never the whole fixed cap up front), at least 8 MB so that the
    // ragged-column launch is always available
    const size_t need = std::max(chase_hip::gemm_f64_ws_need(cplx, opA, m, n, k, num_cu, gemm_min_rounds), (size_t)8 << 20);
    int rc = ensure_ws((need + ((size_t)32 << 20) - 1) & ~(((size_t)32 << 20) - 1));
    if (rc) return rc;
    const int ph = (phase >= 0 && phase <= 3) ? phase : 0;
    int e = chase_hip::gemm_f64(stream, cplx, opA, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, (double*)ws, ws_bytes,
                                num_cu, phase, device, &flops_exec[ph], gemm_min_rounds);
    if (e == chase_hip::GEMM_F64_EWORKSPACE)
        return chase_hip::set_error(CHASE_HIP_EINVAL, "gemm: split-K workspace smaller than gemm_f64_ws_need for this shape");
    if (e) return chase_hip::hip_fail((hipError_t)e, "gemm launch");
    flops_model[ph] += 2.0 * (cplx ? 4.0 : 1.0) * m * (double)n * k;
    ++gemm_calls[ph];
    return 0;
}

void chase_hip_ctx::oplog_add(const char* name, long a, long b, long c, long d)
{
    if (!oplog_on || oplog_mute > 0) return;
    char buf[160];
    snprintf(buf, sizeof buf, "%s %ld %ld %ld %ld", name, a, b, c, d);
    oplog.emplace_back(buf);
}

int chase_hip_ctx::ensure_ws(size_t bytes)
{
    if (ws_bytes >= bytes) return 0;
    (void)hipSetDevice(device);
    if (ws) { (void)hipStreamSynchronize(stream); (void)hipFree(ws); ws = nullptr; ws_bytes = 0; }
    hipError_t e = hipMalloc(&ws, bytes);
    if (e != hipSuccess) return chase_hip::set_error(CHASE_HIP_ENOMEM, "workspace allocation failed");
    ws_bytes = bytes;
    return 0;
}

int chase_hip_ctx::ensure_buf(int idx, size_t bytes)
{
    if (buf_bytes[idx] >= bytes) return 0;
    (void)hipSetDevice(device);
    if (bufs[idx]) { (void)hipStreamSynchronize(stream); (void)hipFree(bufs[idx]); bufs[idx] = nullptr; buf_bytes[idx] = 0; }
    hipError_t e = hipMalloc(&bufs[idx], bytes);
    if (e != hipSuccess) return chase_hip::set_error(CHASE_HIP_ENOMEM, "scratch allocation failed");
    buf_bytes[idx] = bytes;
    return 0;
}

int chase_hip_ctx::ensure_hstage(size_t bytes)
{
    if (hstage_bytes >= bytes) return 0;
    (void)hipSetDevice(device);
    if (hstage) { (void)hipStreamSynchronize(stream); (void)hipHostFree(hstage); hstage = nullptr; hstage_bytes = 0; }
    hipError_t e = hipHostMalloc(&hstage, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return chase_hip::set_error(CHASE_HIP_ENOMEM, "pinned staging allocation failed");
    hstage_bytes = bytes;
    return 0;
}

namespace chase_hip {

__global__ __launch_bounds__(256) void stream_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n16)
{
    // four independent 16-byte loads in flight per lane (round 6: one per lane - 8 MB in flight on the chip - read 4.96 TB/s where
    // the library's own copy2d kernel reaches 6.0; Little's law at ~2 us of loaded latency asks for >= 12 MB)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

int stream_copy(hipStream_t st, void* dst, const void* src, size_t bytes)
{
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(256), 0, st, (float4*)dst, (const float4*)src, n16);
    return (int)hipGetLastError();
}

} // namespace chase_hip

// =====================================================================================================================
// O(N*n) streaming kernels.  All complex data is interleaved (re,im); where the operation acts identically on both
// parts (real scalars) the kernels treat a column as md = m*ept doubles.
// Reductions: one 256-thread workgroup per column, wave64 __shfl_down tree, then 4 partials through LDS.  The
// summation order is fixed, so results are bitwise reproducible run to run (needed for replicated data).
// =====================================================================================================================
namespace chase_hip {

__device__ __forceinline__ double wave_sum(double v)
{
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// all threads of a 256-thread block call this; result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* sm /*[4]*/)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

// H[i,i] += shift (real part for complex).  Reference: cuda/shiftDiagonal.cu:23-50, chase_cpu.hpp:384-389
__global__ void shift_diag_kernel(double* __restrict__ H, long ld, int n, int ept, double shift)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) H[((long)i * ld + i) * ept] += shift;
}

// H[r[i], c[i]] += shift for a list of local (row, col) pairs.  Reference: cuda/shiftDiagonal.cu:100-149 (shift_mgpu)
__global__ void shift_list_kernel(double* __restrict__ H, long ld, const int* __restrict__ rows,
                                  const int* __restrict__ cols, int cnt, int ept, double shift)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) H[((long)cols[i] * ld + rows[i]) * ept] += shift;
}

// A[i,i] += sqrt(m) * eps * (*nrmf) with the shift read from device memory.  Reference: cuda/shiftDiagonal.cu:52-98
__global__ void shift_diag_dev_kernel(double* __restrict__ A, long ld, int n, int ept, const double* __restrict__ nrmf,
                                      double factor)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[((long)i * ld + i) * ept] += factor * nrmf[0];
}

// out[0] = sum_i |A[i,i]|.  Reference: cuda/absTrace.cu:113-167, cpu computeDiagonalAbsSum
__global__ __launch_bounds__(256) void abs_trace_kernel(const double* __restrict__ A, long ld, int n, int ept,
                                                        double* __restrict__ out)
{
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double* p = A + ((long)i * ld + i) * ept;
        s += (ept == 2) ? hypot(p[0], p[1]) : fabs(p[0]);
    }
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) out[0] = s;
}

// strided 2-D copy (lacpy 'A').  Reference: cuda/lacpy.cu:62-530
__global__ __launch_bounds__(256) void copy2d_kernel(const double* __restrict__ src, long lds_, double* __restrict__ dst,
                                                     long ldd, long md, int ncols, int vec2)
{
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
        const double* s = src + (long)j * lds_;
        double* d = dst + (long)j * ldd;
        if (vec2) {
            const long n2 = md >> 1;
            for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
                ((double2*)d)[i] = ((const double2*)s)[i];
        } else {
            for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < md; i += (long)gridDim.x * 256) d[i] = s[i];
        }
    }
}

// swap two columns.  Reference: chase_gpu.hpp:1003-1005 (cublasTswap), chase_cpu.hpp:820-830
__global__ __launch_bounds__(256) void swap_kernel(double* __restrict__ a, double* __restrict__ b, long md)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < md; i += (long)gridDim.x * 256) {
        const double t = a[i]; a[i] = b[i]; b[i] = t;
    }
}

// out[j] = sum_i |W[i,j] - lambda[j] * V[i,j]|^2  (sqrt optional).  V == nullptr -> plain column norms.
// Reference: cuda/residuals.cu:113-296 (one block per column), cpu/residuals.hpp:72-79 (axpy + nrm2)
__global__ __launch_bounds__(256) void resid_norms_kernel(const double* __restrict__ W, long ldw,
                                                          const double* __restrict__ V, long ldv,
                                                          const double* __restrict__ lambda, long md,
                                                          double* __restrict__ out, int do_sqrt)
{
    __shared__ double sm[4];
    const int j = blockIdx.x;
    const double* w = W + (long)j * ldw;
    double s = 0.0;
    if (V) {
        const double* v = V + (long)j * ldv;
        const double lam = lambda[j];
        for (long i = threadIdx.x; i < md; i += 256) { const double r = w[i] - lam * v[i]; s += r * r; }
    } else {
        for (long i = threadIdx.x; i < md; i += 256) { const double r = w[i]; s += r * r; }
    }
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) out[j] = do_sqrt ? sqrt(s) : s;
}

__global__ void sqrt_inplace_kernel(double* __restrict__ x, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = sqrt(x[i]);
}

// out[j] = x_j^H y_j  (complex: (re,im) pair; real: one double).  Reference: cuda/lanczos_kernels.cu batched_dot_product
template <bool CPLX>
__global__ __launch_bounds__(256) void col_dot_kernel(const double* __restrict__ X, long ldx, const double* __restrict__ Y,
                                                      long ldy, int m, double* __restrict__ out)
{
    __shared__ double sm[4];
    const int j = blockIdx.x;
    if constexpr (CPLX) {
        const double2* x = (const double2*)(X + (long)j * ldx);
        const double2* y = (const double2*)(Y + (long)j * ldy);
        double sr = 0.0, si = 0.0;
        for (int i = threadIdx.x; i < m; i += 256) {
            const double2 a = x[i], b = y[i];
            sr += a.x * b.x + a.y * b.y;
            si += a.x * b.y - a.y * b.x;
        }
        sr = block_sum_256(sr, sm);
        si = block_sum_256(si, sm);
        if (threadIdx.x == 0) { out[2 * j] = sr; out[2 * j + 1] = si; }
    } else {
        const double* x = X + (long)j * ldx;
        const double* y = Y + (long)j * ldy;
        double s = 0.0;
        for (int i = threadIdx.x; i < m; i += 256) s += x[i] * y[i];
        s = block_sum_256(s, sm);
        if (threadIdx.x == 0) out[j] = s;
    }
}

// Y_j += sgn * a_j * X_j with a_j read from device memory (complex a for CPLX unless a_is_real).
// Reference: cuda/lanczos_kernels.cu fused_dot_axpy_negate / batched_axpy; batchedAxpyScalar (:161-210) when a_stride==0
template <bool CPLX>
__global__ __launch_bounds__(256) void col_axpy_kernel(const double* __restrict__ a, int a_is_real, int a_stride,
                                                       double sgn, const double* __restrict__ X, long ldx,
                                                       double* __restrict__ Y, long ldy, int m, int ncols)
{
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
        if constexpr (CPLX) {
            double ar, ai;
            if (a_is_real) { ar = sgn * a[(long)j * a_stride]; ai = 0.0; }
            else { ar = sgn * a[2L * j * a_stride]; ai = sgn * a[2L * j * a_stride + 1]; }
            const double2* x = (const double2*)(X + (long)j * ldx);
            double2* y = (double2*)(Y + (long)j * ldy);
            for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
                const double2 xv = x[i]; double2 yv = y[i];
                yv.x += ar * xv.x - ai * xv.y;
                yv.y += ar * xv.y + ai * xv.x;
                y[i] = yv;
            }
        } else {
            const double av = sgn * a[(long)j * a_stride];
            const double* x = X + (long)j * ldx;
            double* y = Y + (long)j * ldy;
            for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) y[i] += av * x[i];
        }
    }
}

// X_j *= (inv ? 1/a_j : a_j), a real, device resident.  Reference: cuda/lanczos_kernels.cu batched_scale / normalize_vectors
__global__ __launch_bounds__(256) void col_scal_kernel(const double* __restrict__ a, int inv, double* __restrict__ X,
                                                       long ldx, long md, int ncols)
{
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
        const double s = inv ? 1.0 / a[j] : a[j];
        double* x = X + (long)j * ldx;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < md; i += (long)gridDim.x * 256) x[i] *= s;
    }
}

// rows >= row0: X[i, j] *= s.  Reference: cuda/flipSign.cu (flipLowerHalfMatrixSign with s = -1, scaleLowerBlockRows)
__global__ __launch_bounds__(256) void scale_rows_kernel(double* __restrict__ X, long ldx, long row0_d, long md,
                                                         int ncols, double s)
{
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
        double* x = X + (long)j * ldx;
        for (long i = row0_d + (long)blockIdx.x * 256 + threadIdx.x; i < md; i += (long)gridDim.x * 256) x[i] *= s;
    }
}

// rows whose GLOBAL index (1D block-cyclic: block nb, p ranks, this rank q) is >= g0: X[i, j] *= s.  The distributed
// flipLowerHalfMatrixSign (linalg/internal/mpi/flipSign.hpp) for block and block-cyclic multivectors alike.
// ept = doubles per element.
__global__ __launch_bounds__(256) void scale_rows_bc_kernel(double* __restrict__ X, long ldx_d, long m, int ncols, int ept,
                                                            long g0, long nb, int p, int q, double s)
{
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
        double* x = X + (long)j * ldx_d;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < m * ept; i += (long)gridDim.x * 256) {
            const long l = i / ept;
            const long g = ((l / nb) * p + q) * nb + l % nb;
            if (g >= g0) x[i] *= s;
        }
    }
}

// in-place complex conjugate.  Reference: cuda/conjugate.cu:21-70
__global__ __launch_bounds__(256) void conj_kernel(double* __restrict__ X, long ldx_d, int m, int ncols)
{
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
        double* x = X + (long)j * ldx_d;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) x[2 * i + 1] = -x[2 * i + 1];
    }
}

// pack / unpack the upper triangle of an n x n matrix into n(n+1)/2 elements (column by column).
// Reference: cuda/lacpy.cu:837-1094 (halves the Gram all-reduce payload, nccl/cholqr.hpp:152-157)
__global__ void pack_upper_kernel(const double* __restrict__ A, long lda, int n, int ept, double* __restrict__ P, int unpack,
                                  double* __restrict__ Aout)
{
    const int j = blockIdx.x;                       // column
    const long base = (long)j * (j + 1) / 2;
    for (int i = threadIdx.x; i <= j; i += blockDim.x) {
        for (int e = 0; e < ept; ++e) {
            if (!unpack) P[(base + i) * ept + e] = A[((long)j * lda + i) * ept + e];
            else Aout[((long)j * lda + i) * ept + e] = P[(base + i) * ept + e];
        }
    }
}

// A[i,j] = conj(A[j,i]) for i > j: rebuild the strictly-lower triangle from the upper one
__global__ void mirror_upper_kernel(double* __restrict__ A, long lda, int n, int ept)
{
    const int j = blockIdx.x;
    for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) {
        const double* s = A + ((long)i * lda + j) * ept;        // A[j,i] (upper)
        double* d = A + ((long)j * lda + i) * ept;              // A[i,j] (lower)
        d[0] = s[0];
        if (ept == 2) d[1] = -s[1];
    }
}

// dst[:, dst_idx[c]] = src[:, src_idx[c]] for c < cnt (batched column gather; applies the deferred Swap() permutation)
__global__ __launch_bounds__(256) void copy_cols_indexed_kernel(const double* __restrict__ src, long lds_, double* __restrict__ dst,
                                                                long ldd, long md, const int* __restrict__ src_idx,
                                                                const int* __restrict__ dst_idx, int cnt)
{
    for (int c = blockIdx.y; c < cnt; c += gridDim.y) {
        const double* s = src + (long)src_idx[c] * lds_;
        double* d = dst + (long)(dst_idx ? dst_idx[c] : c) * ldd;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < md; i += (long)gridDim.x * 256) d[i] = s[i];
    }
}

// row gather / scatter by index list (redistribution between the column-type and row-type multivector layouts):
//   scatter == 0:  out[p, c] = in[idx[p], c]        scatter != 0:  out[idx[p], c] = in[p, c]        p < np, c < ncols
template <int EPT>
__global__ __launch_bounds__(256) void rows_indexed_kernel(const double* __restrict__ in, long ld_in,
                                                           double* __restrict__ out, long ld_out,
                                                           const int* __restrict__ idx, int np, int ncols, int scatter)
{
    for (int c = blockIdx.y; c < ncols; c += gridDim.y) {
        for (int p = blockIdx.x * 256 + threadIdx.x; p < np; p += gridDim.x * 256) {
            const long r = idx[p];
            const double* s = in + ((long)c * ld_in + (scatter ? p : r)) * EPT;
            double* d = out + ((long)c * ld_out + (scatter ? r : p)) * EPT;
            d[0] = s[0];
            if (EPT == 2) d[1] = s[1];
        }
    }
}

// Triangle mask of a block-cyclic shard (distributed symOrHermMatrix, linalg/internal/mpi/symOrHerm.hpp:127-320): local entry
// (il, jl) has the global position (gi, gj) = (block-cyclic row index over pr ranks of block mb, column index over pc ranks of
// block nb); entries of the triangle that is NOT kept become 0, the diagonal is halved.  keep_upper != 0: gi <= gj is kept.
template <int EPT>
__global__ __launch_bounds__(256) void tri_mask_bc_kernel(double* __restrict__ H, long ldh, int mloc, int nloc, long mb, int pr,
                                                          int pi, long nb, int pc, int pj, int keep_upper)
{
    for (int jl = blockIdx.y; jl < nloc; jl += gridDim.y) {
        const long gj = ((long)(jl / nb) * pc + pj) * nb + jl % nb;
        for (int il = blockIdx.x * 256 + threadIdx.x; il < mloc; il += gridDim.x * 256) {
            const long gi = ((long)(il / mb) * pr + pi) * mb + il % mb;
            double* h = H + ((long)jl * ldh + il) * EPT;
            if (gi == gj) { h[0] *= 0.5; if (EPT == 2) h[1] *= 0.5; }
            else if (keep_upper ? (gi > gj) : (gi < gj)) { h[0] = 0.0; if (EPT == 2) h[1] = 0.0; }
        }
    }
}

// H[colmap[b], rowmap[a]] += conj(P[a, b]) for a < nr, b < nc: the conjugate transpose of a packed piece added into a shard
// whose local rows / columns the maps name (second hop of the distributed symOrHermMatrix).  One thread per element of P,
// consecutive threads along a (P is read coalesced; the writes of a wave go to one row of H - strided, but this runs once
// per problem, not per iteration).
template <int EPT>
__global__ __launch_bounds__(256) void conj_transpose_add_kernel(const double* __restrict__ P, long ldp, int nr, int nc,
                                                                 const int* __restrict__ rowmap, const int* __restrict__ colmap,
                                                                 double* __restrict__ H, long ldh)
{
    for (int b = blockIdx.y; b < nc; b += gridDim.y) {
        const long hr = colmap[b];
        for (int a = blockIdx.x * 256 + threadIdx.x; a < nr; a += gridDim.x * 256) {
            const double* p = P + ((long)b * ldp + a) * EPT;
            double* h = H + ((long)rowmap[a] * ldh + hr) * EPT;
            h[0] += p[0];
            if (EPT == 2) h[1] -= p[1];
        }
    }
}

// 64-bit content hash of a strided block of 8-byte words (chase_hip_hash64): every word is mixed with its POSITION (column,
// row) through a 64-bit finaliser and the mixed words are summed modulo 2^64 - integer addition is associative, so the result
// does not depend on the order the atomics land in; two blocks that differ in one bit hash differently with probability
// 1 - 2^-64.  What the multi-rank tests compare instead of downloading the replicas of an eigenvector block.
__global__ __launch_bounds__(256) void hash64_kernel(const unsigned long long* __restrict__ x, long ld_w, long m_w, int ncols,
                                                     unsigned long long* __restrict__ out)
{
    unsigned long long acc = 0;
    for (int c = blockIdx.y; c < ncols; c += gridDim.y)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < m_w; i += (long)gridDim.x * 256) {
            unsigned long long v = x[(long)c * ld_w + i] ^ (0x9E3779B97F4A7C15ull * (unsigned long long)((long)c * m_w + i + 1));
            v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33;
            acc += v;
        }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

// A[i,j] = conj(A[j,i]) for i < j (rebuild the strictly-upper triangle from the lower one) and Im A[j,j] = 0
__global__ void mirror_lower_kernel(double* __restrict__ A, long lda, int n, int ept, int zero_diag_imag)
{
    const int j = blockIdx.x;                                   // column of the lower triangle being read
    if (threadIdx.x == 0 && ept == 2 && zero_diag_imag) A[((long)j * lda + j) * 2 + 1] = 0.0;
    for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) {
        const double* s = A + ((long)j * lda + i) * ept;        // A[i,j] (lower)
        double* d = A + ((long)i * lda + j) * ept;              // A[j,i] (upper)
        d[0] = s[0];
        if (ept == 2) d[1] = -s[1];
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }
static inline dim3 grid2(long md, int ncols)
{
    unsigned gx = cdiv(md, 256 * 4); if (gx < 1) gx = 1; if (gx > 64) gx = 64;
    unsigned gy = ncols < 1 ? 1 : (ncols > 1024 ? 1024 : ncols);
    return dim3(gx, gy);
}

int shift_diag(hipStream_t st, double* H, long ld, int n, int ept, double shift)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(shift_diag_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, H, ld, n, ept, shift);
    return (int)hipGetLastError();
}
int shift_list(hipStream_t st, double* H, long ld, const int* rows, const int* cols, int cnt, int ept, double shift)
{
    if (cnt <= 0) return 0;
    hipLaunchKernelGGL(shift_list_kernel, dim3(cdiv(cnt, 256)), dim3(256), 0, st, H, ld, rows, cols, cnt, ept, shift);
    return (int)hipGetLastError();
}
int shift_diag_dev(hipStream_t st, double* A, long ld, int n, int ept, const double* nrmf, double factor)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(shift_diag_dev_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, A, ld, n, ept, nrmf, factor);
    return (int)hipGetLastError();
}
int abs_trace(hipStream_t st, const double* A, long ld, int n, int ept, double* out_dev)
{
    hipLaunchKernelGGL(abs_trace_kernel, dim3(1), dim3(256), 0, st, A, ld, n, ept, out_dev);
    return (int)hipGetLastError();
}
int copy2d(hipStream_t st, const double* src, long ld_src_d, double* dst, long ld_dst_d, long md, int ncols)
{
    if (md <= 0 || ncols <= 0) return 0;
    const int vec2 = ((md & 1) == 0) && ((ld_src_d & 1) == 0) && ((ld_dst_d & 1) == 0) &&
                     (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 15) == 0);
    hipLaunchKernelGGL(copy2d_kernel, grid2(md, ncols), dim3(256), 0, st, src, ld_src_d, dst, ld_dst_d, md, ncols, vec2);
    return (int)hipGetLastError();
}
int copy_cols_indexed(hipStream_t st, const double* src, long ld_src_d, double* dst, long ld_dst_d, long md,
                      const int* src_idx_dev, const int* dst_idx_dev, int cnt)
{
    if (cnt <= 0 || md <= 0) return 0;
    hipLaunchKernelGGL(copy_cols_indexed_kernel, grid2(md, cnt), dim3(256), 0, st, src, ld_src_d, dst, ld_dst_d, md,
                       src_idx_dev, dst_idx_dev, cnt);
    return (int)hipGetLastError();
}
int copy_cols_indexed_range(hipStream_t st, const double* src, long ld_src_d, double* dst, long ld_dst_d, long md,
                            const int* src_idx_dev, int dst0, int cnt)
{
    if (cnt <= 0 || md <= 0) return 0;
    hipLaunchKernelGGL(copy_cols_indexed_kernel, grid2(md, cnt), dim3(256), 0, st, src, ld_src_d, dst + (long)dst0 * ld_dst_d,
                       ld_dst_d, md, src_idx_dev, (const int*)nullptr, cnt);
    return (int)hipGetLastError();
}
int hash64(hipStream_t st, const double* x, long ld_d, long md, int ncols, unsigned long long* out_dev)
{
    if (md <= 0 || ncols <= 0) return 0;
    hipLaunchKernelGGL(hash64_kernel, grid2(md, ncols), dim3(256), 0, st, (const unsigned long long*)x, ld_d, md, ncols, out_dev);
    return (int)hipGetLastError();
}
int tri_mask_bc(hipStream_t st, bool cplx, double* H, long ldh, int mloc, int nloc, long mb, int pr, int pi, long nb, int pc,
                int pj, int keep_upper)
{
    if (mloc <= 0 || nloc <= 0) return 0;
    const dim3 g = grid2(mloc, nloc);
    if (cplx) hipLaunchKernelGGL(tri_mask_bc_kernel<2>, g, dim3(256), 0, st, H, ldh, mloc, nloc, mb, pr, pi, nb, pc, pj, keep_upper);
    else      hipLaunchKernelGGL(tri_mask_bc_kernel<1>, g, dim3(256), 0, st, H, ldh, mloc, nloc, mb, pr, pi, nb, pc, pj, keep_upper);
    return (int)hipGetLastError();
}
int conj_transpose_add(hipStream_t st, bool cplx, const double* P, long ldp, int nr, int nc, const int* rowmap_dev,
                       const int* colmap_dev, double* H, long ldh)
{
    if (nr <= 0 || nc <= 0) return 0;
    const dim3 g = grid2(nr, nc);
    if (cplx) hipLaunchKernelGGL(conj_transpose_add_kernel<2>, g, dim3(256), 0, st, P, ldp, nr, nc, rowmap_dev, colmap_dev, H, ldh);
    else      hipLaunchKernelGGL(conj_transpose_add_kernel<1>, g, dim3(256), 0, st, P, ldp, nr, nc, rowmap_dev, colmap_dev, H, ldh);
    return (int)hipGetLastError();
}
int rows_indexed(hipStream_t st, bool cplx, const double* in, long ld_in, double* out, long ld_out, const int* idx_dev,
                 int np, int ncols, int scatter)
{
    if (np <= 0 || ncols <= 0) return 0;
    const dim3 g = grid2(np, ncols);
    if (cplx) hipLaunchKernelGGL(rows_indexed_kernel<2>, g, dim3(256), 0, st, in, ld_in, out, ld_out, idx_dev, np, ncols, scatter);
    else      hipLaunchKernelGGL(rows_indexed_kernel<1>, g, dim3(256), 0, st, in, ld_in, out, ld_out, idx_dev, np, ncols, scatter);
    return (int)hipGetLastError();
}
int swap_cols(hipStream_t st, double* a, double* b, long md)
{
    if (md <= 0 || a == b) return 0;
    unsigned g = cdiv(md, 256 * 4); if (g > 256) g = 256;
    hipLaunchKernelGGL(swap_kernel, dim3(g), dim3(256), 0, st, a, b, md);
    return (int)hipGetLastError();
}
int resid_norms(hipStream_t st, const double* W, long ldw_d, const double* V, long ldv_d, const double* lambda_dev,
                long md, int ncols, double* out_dev, int do_sqrt)
{
    if (ncols <= 0) return 0;
    hipLaunchKernelGGL(resid_norms_kernel, dim3(ncols), dim3(256), 0, st, W, ldw_d, V, ldv_d, lambda_dev, md, out_dev,
                       do_sqrt);
    return (int)hipGetLastError();
}
int sqrt_inplace(hipStream_t st, double* x, int n)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sqrt_inplace_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, n);
    return (int)hipGetLastError();
}
int col_dot(hipStream_t st, bool cplx, const double* X, long ldx_d, const double* Y, long ldy_d, int m, int ncols,
            double* out_dev)
{
    if (ncols <= 0) return 0;
    if (cplx) hipLaunchKernelGGL(col_dot_kernel<true>, dim3(ncols), dim3(256), 0, st, X, ldx_d, Y, ldy_d, m, out_dev);
    else      hipLaunchKernelGGL(col_dot_kernel<false>, dim3(ncols), dim3(256), 0, st, X, ldx_d, Y, ldy_d, m, out_dev);
    return (int)hipGetLastError();
}
int col_axpy(hipStream_t st, bool cplx, const double* a_dev, int a_is_real, int a_stride, double sgn, const double* X,
             long ldx_d, double* Y, long ldy_d, int m, int ncols)
{
    if (ncols <= 0 || m <= 0) return 0;
    const dim3 g = grid2(m, ncols);
    if (cplx) hipLaunchKernelGGL(col_axpy_kernel<true>, g, dim3(256), 0, st, a_dev, a_is_real, a_stride, sgn, X, ldx_d, Y, ldy_d, m, ncols);
    else      hipLaunchKernelGGL(col_axpy_kernel<false>, g, dim3(256), 0, st, a_dev, a_is_real, a_stride, sgn, X, ldx_d, Y, ldy_d, m, ncols);
    return (int)hipGetLastError();
}
int col_scal(hipStream_t st, const double* a_dev, int inv, double* X, long ldx_d, long md, int ncols)
{
    if (ncols <= 0 || md <= 0) return 0;
    hipLaunchKernelGGL(col_scal_kernel, grid2(md, ncols), dim3(256), 0, st, a_dev, inv, X, ldx_d, md, ncols);
    return (int)hipGetLastError();
}
int scale_rows(hipStream_t st, double* X, long ldx_d, long row0_d, long md, int ncols, double s)
{
    if (ncols <= 0 || md <= row0_d) return 0;
    hipLaunchKernelGGL(scale_rows_kernel, grid2(md - row0_d, ncols), dim3(256), 0, st, X, ldx_d, row0_d, md, ncols, s);
    return (int)hipGetLastError();
}
int scale_rows_bc(hipStream_t st, double* X, long ldx_d, long m, int ncols, int ept, long g0, long nb, int p, int q, double s)
{
    if (ncols <= 0 || m <= 0) return 0;
    hipLaunchKernelGGL(scale_rows_bc_kernel, grid2(m * ept, ncols), dim3(256), 0, st, X, ldx_d, m, ncols, ept, g0, nb, p, q, s);
    return (int)hipGetLastError();
}
int conj_inplace(hipStream_t st, double* X, long ldx_d, int m, int ncols)
{
    if (ncols <= 0 || m <= 0) return 0;
    hipLaunchKernelGGL(conj_kernel, grid2(m, ncols), dim3(256), 0, st, X, ldx_d, m, ncols);
    return (int)hipGetLastError();
}
int pack_upper(hipStream_t st, const double* A, long lda, int n, int ept, double* P)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pack_upper_kernel, dim3(n), dim3(128), 0, st, A, lda, n, ept, P, 0, (double*)nullptr);
    return (int)hipGetLastError();
}
int unpack_upper(hipStream_t st, double* P, int n, int ept, double* A, long lda)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pack_upper_kernel, dim3(n), dim3(128), 0, st, (const double*)nullptr, lda, n, ept, P, 1, A);
    return (int)hipGetLastError();
}
int mirror_lower(hipStream_t st, double* A, long lda, int n, int ept, int zero_diag_imag)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mirror_lower_kernel, dim3(n), dim3(128), 0, st, A, lda, n, ept, zero_diag_imag);
    return (int)hipGetLastError();
}
int mirror_upper(hipStream_t st, double* A, long lda, int n, int ept)
{
    if (n <= 1) return 0;
    hipLaunchKernelGGL(mirror_upper_kernel, dim3(n), dim3(128), 0, st, A, lda, n, ept);
    return (int)hipGetLastError();
}

} // namespace chase_hip
