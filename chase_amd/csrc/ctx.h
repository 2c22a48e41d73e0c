// ctx.h — the opaque context behind chase_hip_ctx* (include/chase_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string>
#include <vector>

struct chase_hip_ctx {
    int device = 0;
    int num_cu = 0;
    int clock_khz = 0;
    size_t hbm_bytes = 0;
    char name[128] = {0};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int phase = 0;               // 1 between FilterPhaseStart/End: selects the filter-tagged GEMM symbol
    // GEMM accounting per phase (0 other, 1 filter, 2 H-times-block outside the filter, 3 verification products: always
    // four multiplications): flops of the reference's model (2*F*m*n*k, F = 4 complex) and flops the matrix cores executed
    // (3/4 of it for three-multiplication launches)
    double flops_model[4] = {0, 0, 0, 0}, flops_exec[4] = {0, 0, 0, 0};
    unsigned long long gemm_calls[4] = {0, 0, 0, 0};
    int gemm_min_rounds = 0;          // > 0: products share the chip with a collective (chase_hip_ctx_set_gemm_min_rounds)
    void* ws = nullptr;          // split-K slabs, grown on demand
    size_t ws_bytes = 0;
    enum { BUF_TINV = 0, BUF_PANEL, BUF_SCAL, BUF_LAMBDA, BUF_RR, BUF_EIG, BUF_STEDC, BUF_APPLYQ, NBUF };
    void* bufs[NBUF] = {};   // device scratch, grown on demand (BUF_EIG / BUF_STEDC: the projected eigensolver's blocks)
    size_t buf_bytes[NBUF] = {};
    void* hstage = nullptr;      // pinned host staging (HEEVD round trip)
    size_t hstage_bytes = 0;

    // operator log (chase_hip_ctx_oplog): one line per C-ABI operator this context executes - name and shapes, no pointers, no
    // scalars - so that two runs can be compared launch for launch (single-rank replay against the real rank, tests).
    // oplog_mute > 0: inside an operator whose inner launches depend on the DATA (the divide & conquer eigensolver's
    // deflation), only the operator itself is listed.
    bool oplog_on = false;
    int oplog_mute = 0;
    std::vector<std::string> oplog;
    std::string oplog_text;
    void oplog_add(const char* name, long a, long b, long c, long d);

    int ensure_ws(size_t bytes);
    int ensure_buf(int idx, size_t bytes);
    int ensure_hstage(size_t bytes);
    // one GEMM through the MFMA kernel on this context's stream: sizes the split-K workspace for the shape, keeps the books
    int gemm(bool cplx, char opA, int m, int n, int k, const double* alpha, const double* A, long lda, const double* B,
             long ldb, const double* beta, double* C, long ldc);
};

namespace chase_hip {
int set_error(int code, const char* what);
int hip_fail(hipError_t e, const char* where);
}
