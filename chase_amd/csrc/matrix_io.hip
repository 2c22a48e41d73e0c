// matrix_io.hip — raw column-major binary matrix files <-> device shards (the data format either side of the path).
//
// Reference: linalg/matrix/matrix.hpp:313-360 (readFromBinaryFile: N x N elements of T, column-major, no header; the file
// may be larger than needed, smaller is an error), linalg/distMatrix/distMatrix.hpp:2425-2520 (block layout: the rank's
// sub-array, MPI-IO subarray view or seekg per column) and :3210-3330 (block-cyclic: MPI darray view).  Here every rank
// reads only the byte ranges of its own (mb x nb block-cyclic) shard with pread — one contiguous run per (local column,
// row block) — through a pinned staging buffer, and uploads panels of columns with one strided copy each.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>
#include "ctx.h"
#include "kernels.h"
#include "../../include/chase_hip.h"

using namespace chase_hip;

namespace {
struct Fd {
    int fd = -1;
    ~Fd() { if (fd >= 0) ::close(fd); }
};
int io_error(const char* what, const char* path)
{
    std::string m = std::string(what) + " '" + (path ? path : "") + "': " + std::strerror(errno);
    return set_error(CHASE_HIP_EIO, m.c_str());
}
bool full_pread(int fd, char* dst, size_t bytes, off_t off)
{
    while (bytes) {
        const ssize_t r = ::pread(fd, dst, bytes, off);
        if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; }
        dst += r; off += r; bytes -= (size_t)r;
    }
    return true;
}
bool full_pwrite(int fd, const char* src, size_t bytes, off_t off)
{
    while (bytes) {
        const ssize_t r = ::pwrite(fd, src, bytes, off);
        if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; }
        src += r; off += r; bytes -= (size_t)r;
    }
    return true;
}
} // namespace

extern "C" {

int chase_hip_load_matrix_shard(chase_hip_ctx* c, const char* path, int cplx, long N, int mloc, int nloc, int mb, int pr,
                                int pi, int nb, int pc, int pj, void* dev, long ldd)
{
    if (!c || !path || (!dev && mloc > 0 && nloc > 0)) return set_error(CHASE_HIP_EINVAL, "load_matrix_shard: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (N <= 0 || mloc < 0 || nloc < 0 || mb <= 0 || nb <= 0 || pr <= 0 || pc <= 0 || pi < 0 || pi >= pr || pj < 0 ||
        pj >= pc || ldd < mloc)
        return set_error(CHASE_HIP_EINVAL, "load_matrix_shard: bad shape");
    const size_t es = cplx ? 16 : 8;
    Fd f;
    f.fd = ::open(path, O_RDONLY);
    if (f.fd < 0) return io_error("load_matrix_shard: cannot open", path);
    struct stat sb;
    if (::fstat(f.fd, &sb) != 0) return io_error("load_matrix_shard: cannot stat", path);
    if ((unsigned long long)sb.st_size < (unsigned long long)N * (unsigned long long)N * es)
        return set_error(CHASE_HIP_EIO, "load_matrix_shard: file is smaller than the N x N matrix");
    if (mloc == 0 || nloc == 0) return 0;
    // the last local row / column must exist in the global matrix
    const long g_last_r = ((long)((mloc - 1) / mb) * pr + pi) * mb + (mloc - 1) % mb;
    const long g_last_c = ((long)((nloc - 1) / nb) * pc + pj) * nb + (nloc - 1) % nb;
    if (g_last_r >= N || g_last_c >= N) return set_error(CHASE_HIP_EINVAL, "load_matrix_shard: shard exceeds the matrix");
    // panels of columns through the pinned staging buffer (<= 256 MB)
    const size_t col_bytes = (size_t)mloc * es;
    int pcols = (int)std::max<size_t>(1, std::min<size_t>((size_t)nloc, ((size_t)256 << 20) / col_bytes));
    int rc = c->ensure_hstage((size_t)pcols * col_bytes);
    if (rc) return rc;
    char* stage = (char*)c->hstage;
    for (int j0 = 0; j0 < nloc; j0 += pcols) {
        const int w = std::min(pcols, nloc - j0);
        if (hipStreamSynchronize(c->stream) != hipSuccess) return set_error(CHASE_HIP_EIO, "load_matrix_shard: sync failed");
        for (int j = 0; j < w; ++j) {
            const int lj = j0 + j;
            const long gj = ((long)(lj / nb) * pc + pj) * nb + lj % nb;
            for (int i0 = 0; i0 < mloc; i0 += mb) {                       // one contiguous run of the file per row block
                const int h = std::min(mb, mloc - i0);
                const long gi = ((long)(i0 / mb) * pr + pi) * mb;
                const off_t off = (off_t)(((unsigned long long)gj * (unsigned long long)N + (unsigned long long)gi) * es);
                if (!full_pread(f.fd, stage + ((size_t)j * mloc + i0) * es, (size_t)h * es, off))
                    return io_error("load_matrix_shard: short read from", path);
            }
        }
        hipError_t e = hipMemcpy2DAsync((char*)dev + (size_t)j0 * ldd * es, (size_t)ldd * es, stage, col_bytes, col_bytes,
                                        (size_t)w, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return hip_fail(e, "load_matrix_shard: upload");
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return set_error(CHASE_HIP_EIO, "load_matrix_shard: sync failed");
    return 0;
}

int chase_hip_save_matrix(chase_hip_ctx* c, const char* path, int cplx, int m, int n, const void* dev, long ldd)
{
    if (!c || !path || (!dev && m > 0 && n > 0)) return set_error(CHASE_HIP_EINVAL, "save_matrix: NULL argument");
    (void)hipSetDevice(c->device);      // entry points may be called with another device current
    if (m < 0 || n < 0 || ldd < m) return set_error(CHASE_HIP_EINVAL, "save_matrix: bad shape");
    const size_t es = cplx ? 16 : 8;
    Fd f;
    f.fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (f.fd < 0) return io_error("save_matrix: cannot open", path);
    if (m == 0 || n == 0) return 0;
    const size_t col_bytes = (size_t)m * es;
    int pcols = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)256 << 20) / col_bytes));
    int rc = c->ensure_hstage((size_t)pcols * col_bytes);
    if (rc) return rc;
    for (int j0 = 0; j0 < n; j0 += pcols) {
        const int w = std::min(pcols, n - j0);
        hipError_t e = hipMemcpy2DAsync(c->hstage, col_bytes, (const char*)dev + (size_t)j0 * ldd * es, (size_t)ldd * es,
                                        col_bytes, (size_t)w, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) return hip_fail(e, "save_matrix: download");
        if (hipStreamSynchronize(c->stream) != hipSuccess) return set_error(CHASE_HIP_EIO, "save_matrix: sync failed");
        if (!full_pwrite(f.fd, (const char*)c->hstage, (size_t)w * col_bytes, (off_t)((size_t)j0 * col_bytes)))
            return io_error("save_matrix: short write to", path);
    }
    return 0;
}

/* the counterpart of chase_hip_load_matrix_shard: this rank's block-cyclic shard goes to its byte ranges of the shared
 * N x N file (one pwrite per local column and row block).  The file is created if missing and never truncated, so the
 * ranks of a grid may call this concurrently on the same path (the reference writes through an MPI-IO darray / subarray
 * view, linalg/distMatrix/distMatrix.hpp:2241-2300,3117-3200). */
int chase_hip_save_matrix_shard(chase_hip_ctx* c, const char* path, int cplx, long N, int mloc, int nloc, int mb, int pr,
                                int pi, int nb, int pc, int pj, const void* dev, long ldd)
{
    if (!c || !path || (!dev && mloc > 0 && nloc > 0)) return set_error(CHASE_HIP_EINVAL, "save_matrix_shard: NULL argument");
    (void)hipSetDevice(c->device);
    if (N <= 0 || mloc < 0 || nloc < 0 || mb <= 0 || nb <= 0 || pr <= 0 || pc <= 0 || pi < 0 || pi >= pr || pj < 0 ||
        pj >= pc || ldd < mloc)
        return set_error(CHASE_HIP_EINVAL, "save_matrix_shard: bad shape");
    const size_t es = cplx ? 16 : 8;
    Fd f;
    f.fd = ::open(path, O_WRONLY | O_CREAT, 0644);
    if (f.fd < 0) return io_error("save_matrix_shard: cannot open", path);
    if (mloc == 0 || nloc == 0) return 0;
    const long g_last_r = ((long)((mloc - 1) / mb) * pr + pi) * mb + (mloc - 1) % mb;
    const long g_last_c = ((long)((nloc - 1) / nb) * pc + pj) * nb + (nloc - 1) % nb;
    if (g_last_r >= N || g_last_c >= N) return set_error(CHASE_HIP_EINVAL, "save_matrix_shard: shard exceeds the matrix");
    const size_t col_bytes = (size_t)mloc * es;
    int pcols = (int)std::max<size_t>(1, std::min<size_t>((size_t)nloc, ((size_t)256 << 20) / col_bytes));
    int rc = c->ensure_hstage((size_t)pcols * col_bytes);
    if (rc) return rc;
    const char* stage = (const char*)c->hstage;
    for (int j0 = 0; j0 < nloc; j0 += pcols) {
        const int w = std::min(pcols, nloc - j0);
        hipError_t e = hipMemcpy2DAsync(c->hstage, col_bytes, (const char*)dev + (size_t)j0 * ldd * es, (size_t)ldd * es,
                                        col_bytes, (size_t)w, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) return hip_fail(e, "save_matrix_shard: download");
        if (hipStreamSynchronize(c->stream) != hipSuccess) return set_error(CHASE_HIP_EIO, "save_matrix_shard: sync failed");
        for (int j = 0; j < w; ++j) {
            const int lj = j0 + j;
            const long gj = ((long)(lj / nb) * pc + pj) * nb + lj % nb;
            for (int i0 = 0; i0 < mloc; i0 += mb) {
                const int h = std::min(mb, mloc - i0);
                const long gi = ((long)(i0 / mb) * pr + pi) * mb;
                const off_t off = (off_t)(((unsigned long long)gj * (unsigned long long)N + (unsigned long long)gi) * es);
                if (!full_pwrite(f.fd, stage + ((size_t)j * mloc + i0) * es, (size_t)h * es, off))
                    return io_error("save_matrix_shard: short write to", path);
            }
        }
    }
    return 0;
}

} // extern "C"
