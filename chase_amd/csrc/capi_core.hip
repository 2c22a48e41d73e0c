// capi_core.hip — context, memory plumbing, GEMM and probe entry points of the C ABI (include/chase_hip.h)
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/chase_hip.h"
#include "ctx.h"
#include "kernels.h"

namespace chase_hip {
thread_local std::string g_last_error;
int set_error(int code, const char* what)
{
    g_last_error = what ? what : "";
    return code;
}
int hip_fail(hipError_t e, const char* where)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s", where, hipGetErrorString(e));
    g_last_error = buf;
    return -(int)e;
}
} // namespace chase_hip

using namespace chase_hip;

#define HIPCHK(x)                                                                                                      \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) return hip_fail(e_, #x);                                                                 \
    } while (0)

extern "C" {

const char* chase_hip_version(void) { return "chase_hip 0.1 (gfx950)"; }
int chase_hip_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
const char* chase_hip_last_error(void) { return g_last_error.c_str(); }

int chase_hip_ctx_create(chase_hip_ctx** out, int device, void* stream)
{
    if (!out) return set_error(CHASE_HIP_EINVAL, "ctx_create: out == NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return set_error(CHASE_HIP_ENODEV, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_error(CHASE_HIP_EINVAL, "ctx_create: bad device ordinal");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        char buf[256];
        snprintf(buf, sizeof buf, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device,
                 prop.gcnArchName);
        return set_error(CHASE_HIP_ENODEV, buf);
    }
    chase_hip_ctx* c = new chase_hip_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    c->clock_khz = prop.clockRate;
    c->hbm_bytes = prop.totalGlobalMem;
    snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    *out = c;
    return 0;
}

int chase_hip_ctx_destroy(chase_hip_ctx* c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->ws) (void)hipFree(c->ws);
    for (int i = 0; i < chase_hip_ctx::NBUF; ++i)
        if (c->bufs[i]) (void)hipFree(c->bufs[i]);
    if (c->hstage) (void)hipHostFree(c->hstage);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int chase_hip_ctx_sync(chase_hip_ctx* c)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

void* chase_hip_ctx_stream(chase_hip_ctx* c) { return c ? (void*)c->stream : nullptr; }

/* operator log: on != 0 starts a fresh log, 0 stops it; the text ('\n'-separated lines "name a b c d") stays readable until
 * the next call with on != 0 */
int chase_hip_ctx_oplog(chase_hip_ctx* c, int on)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    if (on) { c->oplog.clear(); c->oplog_mute = 0; }
    c->oplog_on = on != 0;
    return 0;
}
/* delta > 0: operators executed from now on are not listed (nestable), delta < 0: listed again */
int chase_hip_ctx_oplog_mute(chase_hip_ctx* c, int delta)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    c->oplog_mute += delta;
    if (c->oplog_mute < 0) c->oplog_mute = 0;
    return 0;
}
const char* chase_hip_ctx_oplog_text(chase_hip_ctx* c)
{
    if (!c) return "";
    c->oplog_text.clear();
    for (const auto& l : c->oplog) { c->oplog_text += l; c->oplog_text += '\n'; }
    return c->oplog_text.c_str();
}

int chase_hip_device_bus_id(chase_hip_ctx* c, char* out, int len)
{
    if (!c || !out || len < 16) return set_error(CHASE_HIP_EINVAL, "device_bus_id: bad argument");
    HIPCHK(hipDeviceGetPCIBusId(out, len, c->device));
    return 0;
}

int chase_hip_device_info(chase_hip_ctx* c, int* num_cu, int* clock_khz, size_t* hbm_bytes, char* name, int name_len)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (num_cu) *num_cu = c->num_cu;
    if (clock_khz) *clock_khz = c->clock_khz;
    if (hbm_bytes) *hbm_bytes = c->hbm_bytes;
    if (name && name_len > 0) {
        strncpy(name, c->name, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    return 0;
}

int chase_hip_malloc(chase_hip_ctx* c, void** dev, size_t bytes)
{
    if (!c || !dev) return set_error(CHASE_HIP_EINVAL, "malloc: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    HIPCHK(hipSetDevice(c->device));
    hipError_t e = hipMalloc(dev, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) return set_error(CHASE_HIP_ENOMEM, "hipMalloc: out of memory");
    HIPCHK(e);
    return 0;
}
int chase_hip_free(chase_hip_ctx* c, void* dev)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (dev) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipFree(dev));
    }
    return 0;
}
int chase_hip_memcpy_h2d(chase_hip_ctx* c, void* dev, const void* host, size_t bytes)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    HIPCHK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int chase_hip_memcpy_d2h(chase_hip_ctx* c, void* host, const void* dev, size_t bytes)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int chase_hip_memcpy_d2d(chase_hip_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int chase_hip_memset(chase_hip_ctx* c, void* dev, int value, size_t bytes)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    HIPCHK(hipMemsetAsync(dev, value, bytes, c->stream));
    return 0;
}
int chase_hip_timer_start(chase_hip_ctx* c)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    return 0;
}
int chase_hip_timer_stop(chase_hip_ctx* c, float* ms)
{
    if (!c || !ms) return set_error(CHASE_HIP_EINVAL, "timer_stop: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}

static int check_gemm(char opA, int m, int n, int k, const void* A, long lda, const void* B, long ldb, const void* C,
                      long ldc)
{
    const bool opn = (opA == 'N' || opA == 'n');
    const bool opc = (opA == 'C' || opA == 'c' || opA == 'T' || opA == 't');
    if (!opn && !opc) return set_error(CHASE_HIP_EINVAL, "gemm: opA must be 'N' or 'C'");
    if (m < 0 || n < 0 || k < 0) return set_error(CHASE_HIP_EINVAL, "gemm: negative dimension");
    if (m == 0 || n == 0) return 0;
    if (!C || (k > 0 && (!A || !B))) return set_error(CHASE_HIP_EINVAL, "gemm: NULL matrix pointer");
    const long arows = opn ? m : k;
    // (an operand without rows - k = 0: a rank that owns no rows of the block - may come with leading dimension 0; the
    // product is then beta C and nothing of A or B is read)
    if (lda < arows || ldb < k || ldc < m || (k > 0 && (lda < 1 || ldb < 1)))
        return set_error(CHASE_HIP_EINVAL, "gemm: leading dimension too small");
    return 0;
}

int chase_hip_gemm_d(chase_hip_ctx* c, char opA, int m, int n, int k, double alpha, const double* A, long lda,
                     const double* B, long ldb, double beta, double* C, long ldc)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    int rc = check_gemm(opA, m, n, k, A, lda, B, ldb, C, ldc);
    if (rc) return rc;
    return c->gemm(false, opA, m, n, k, &alpha, A, lda, B, ldb, &beta, C, ldc);
}

int chase_hip_gemm_z(chase_hip_ctx* c, char opA, int m, int n, int k, const double alpha[2], const void* A, long lda,
                     const void* B, long ldb, const double beta[2], void* C, long ldc)
{
    if (!c) return set_error(CHASE_HIP_EINVAL, "ctx == NULL");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (!alpha || !beta) return set_error(CHASE_HIP_EINVAL, "gemm_z: NULL alpha/beta");
    int rc = check_gemm(opA, m, n, k, A, lda, B, ldb, C, ldc);
    if (rc) return rc;
    return c->gemm(true, opA, m, n, k, alpha, (const double*)A, lda, (const double*)B, ldb, beta, (double*)C, ldc);
}

/* bytes of split-K workspace chase_hip_gemm_{d,z} uses for this shape on a device with num_cu compute units (a pure function
 * of the shape: the launcher never picks another split to fit what happens to be allocated) */
size_t chase_hip_gemm_workspace_bytes(int cplx, char opA, int m, int n, int k, int num_cu, int min_rounds)
{
    return gemm_f64_ws_need(cplx != 0, opA, m, n, k, num_cu, min_rounds);
}
int chase_hip_gemm3m_enabled(void) { return gemm3m_enabled(); }
int chase_hip_set_gemm3m(int on)
{
    gemm3m_set(on);
    return 0;
}

int chase_hip_ctx_gemm_counters(chase_hip_ctx* c, int phase, double* flops_model, double* flops_executed,
                                unsigned long long* calls, int reset)
{
    if (!c || phase < 0 || phase > 3) return set_error(CHASE_HIP_EINVAL, "gemm_counters: bad argument");
    if (flops_model) *flops_model = c->flops_model[phase];
    if (flops_executed) *flops_executed = c->flops_exec[phase];
    if (calls) *calls = c->gemm_calls[phase];
    if (reset) { c->flops_model[phase] = c->flops_exec[phase] = 0; c->gemm_calls[phase] = 0; }
    return 0;
}

int chase_hip_mfma_f64_peak(chase_hip_ctx* c, double* tflops)
{
    if (!c || !tflops) return set_error(CHASE_HIP_EINVAL, "mfma_peak: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    const int blocks = c->num_cu * 2, iters = 16384;           // ~7 ms at peak: long enough for the clock to settle
    double* out = nullptr;
    HIPCHK(hipMalloc((void**)&out, (size_t)blocks * 256 * sizeof(double)));
    mfma_f64_peak(c->stream, out, blocks, 64); // warm-up
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        int e = mfma_f64_peak(c->stream, out, blocks, iters);
        if (e) { hipFree(out); return hip_fail((hipError_t)e, "mfma_peak launch"); }
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        HIPCHK(hipEventSynchronize(c->ev1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        if (ms < best) best = ms;
    }
    HIPCHK(hipFree(out));
    // per wave per iteration: 16 MFMAs x (16*16*4*2) flops
    const double flops = (double)blocks * 4 /*waves*/ * iters * 16.0 * 2048.0;
    *tflops = flops / (best * 1e-3) / 1e12;
    return 0;
}

int chase_hip_hbm_copy_peak(chase_hip_ctx* c, size_t bytes, double* gbps)
{
    if (!c || !gbps) return set_error(CHASE_HIP_EINVAL, "hbm_copy_peak: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    bytes &= ~(size_t)4095;
    if (bytes < 4096) return set_error(CHASE_HIP_EINVAL, "hbm_copy_peak: size too small");
    char *a = nullptr, *b = nullptr;
    HIPCHK(hipMalloc((void**)&a, bytes));
    hipError_t e = hipMalloc((void**)&b, bytes);
    if (e != hipSuccess) { hipFree(a); return hip_fail(e, "hipMalloc"); }
    hipMemsetAsync(a, 1, bytes, c->stream);
    stream_copy(c->stream, b, a, bytes);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(c->ev0, c->stream);
        stream_copy(c->stream, b, a, bytes);
        hipEventRecord(c->ev1, c->stream);
        hipEventSynchronize(c->ev1);
        float ms = 0;
        hipEventElapsedTime(&ms, c->ev0, c->ev1);
        if (ms < best) best = ms;
    }
    hipFree(a);
    hipFree(b);
    *gbps = 2.0 * (double)bytes / (best * 1e-3) / 1e9;
    return 0;
}

} // extern "C"
