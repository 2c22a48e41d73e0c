// grid.hip — 2D process grid + collectives (RCCL over xGMI, or host-callback test transport).  See chase_hip_grid.h.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "grid.h"
#include "kernels.h"

using namespace chase_hip;

#define HIPCHK(x)                                                                                                      \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) return hip_fail(e_, #x);                                                                 \
    } while (0)
#define NCCLCHK(x)                                                                                                     \
    do {                                                                                                               \
        ncclResult_t r_ = (x);                                                                                         \
        if (r_ != ncclSuccess) {                                                                                       \
            char b_[256];                                                                                              \
            snprintf(b_, sizeof b_, "%s: %s", #x, ncclGetErrorString(r_));                                            \
            return set_error(CHASE_HIP_ECOMM, b_);                                                                     \
        }                                                                                                              \
    } while (0)

static int grid_common(chase_hip_grid* g, chase_hip_ctx* ctx, int nprow, int npcol, int rank)
{
    if (!ctx) return set_error(CHASE_HIP_EINVAL, "grid: NULL ctx");
    if (nprow < 1 || npcol < 1 || rank < 0 || rank >= nprow * npcol)
        return set_error(CHASE_HIP_EINVAL, "grid: bad dimensions / rank");
    g->ctx = ctx; g->nprow = nprow; g->npcol = npcol; g->rank = rank;
    g->myrow = rank % nprow;                       // column-major grid ordering (grid/mpiGrid2D.hpp:402-432)
    g->mycol = rank / nprow;
    HIPCHK(hipSetDevice(ctx->device));
    // the collectives' few workgroups must not queue behind a chip-filling GEMM launch: highest stream priority
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIPCHK(hipStreamCreateWithPriority(&g->comm_stream, hipStreamNonBlocking, prio_hi));
    HIPCHK(hipEventCreateWithFlags(&g->ev_compute, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&g->ev_comm, hipEventDisableTiming));
    HIPCHK(hipMalloc((void**)&g->scal_dev, 64));
    return 0;
}

extern "C" {

int chase_hip_rccl_unique_id(char id[CHASE_HIP_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == CHASE_HIP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return 0;
}

int chase_hip_grid_create_rccl(chase_hip_grid** out, chase_hip_ctx* ctx, int nprow, int npcol, int rank,
                               const char* id_row, const char* id_col)
{
    if (!out) return set_error(CHASE_HIP_EINVAL, "grid_create: NULL out");
    chase_hip_grid* g = new chase_hip_grid();
    int rc = grid_common(g, ctx, nprow, npcol, rank);
    if (rc) { delete g; return rc; }
    g->use_rccl = true;
    g->force = getenv("CHASE_HIP_RCCL_FORCE") != nullptr;
    // the reference creates its row and column NCCL communicators the same way: one unique id per sub-communicator,
    // ncclCommInitRank on each (grid/mpiGrid2D.hpp:448-484)
    if (npcol > 1 || g->force) {
        if (!id_row) { delete g; return set_error(CHASE_HIP_EINVAL, "grid_create: row id missing"); }
        ncclUniqueId u; memcpy(&u, id_row, sizeof u);
        NCCLCHK(ncclCommInitRank(&g->comm[CHASE_HIP_ROW], npcol, u, g->mycol));
    }
    if (nprow > 1 || g->force) {
        if (!id_col) { delete g; return set_error(CHASE_HIP_EINVAL, "grid_create: col id missing"); }
        ncclUniqueId u; memcpy(&u, id_col, sizeof u);
        NCCLCHK(ncclCommInitRank(&g->comm[CHASE_HIP_COL], nprow, u, g->myrow));
    }
    // first collective on a communicator sets up the xGMI connections (hundreds of ms): pay it here, not in the first
    // filter step, and surface transport problems at construction
    HIPCHK(hipMemsetAsync(g->scal_dev, 0, 64, g->comm_stream));
    for (int grp = 0; grp < 2; ++grp)
        if (g->comm[grp]) NCCLCHK(ncclAllReduce(g->scal_dev, g->scal_dev, 8, ncclDouble, ncclSum, g->comm[grp], g->comm_stream));
    HIPCHK(hipStreamSynchronize(g->comm_stream));
    *out = g;
    return 0;
}

int chase_hip_grid_create_host(chase_hip_grid** out, chase_hip_ctx* ctx, int nprow, int npcol, int rank,
                               chase_hip_host_allreduce_fn allreduce, chase_hip_host_bcast_fn bcast, void* user)
{
    if (!out) return set_error(CHASE_HIP_EINVAL, "grid_create: NULL out");
    if (nprow * npcol > 1 && (!allreduce || !bcast)) return set_error(CHASE_HIP_EINVAL, "grid_create: NULL callback");
    chase_hip_grid* g = new chase_hip_grid();
    int rc = grid_common(g, ctx, nprow, npcol, rank);
    if (rc) { delete g; return rc; }
    g->use_rccl = false;
    g->h_allreduce = allreduce; g->h_bcast = bcast; g->h_user = user;
    *out = g;
    return 0;
}

int chase_hip_grid_destroy(chase_hip_grid* g)
{
    if (!g) return 0;
    hipSetDevice(g->ctx->device);
    hipStreamSynchronize(g->comm_stream);
    for (int i = 0; i < 2; ++i)
        if (g->comm[i]) ncclCommDestroy(g->comm[i]);
    hipFree(g->scal_dev);
    for (hipEvent_t e : g->slots)
        if (e) hipEventDestroy(e);
    hipEventDestroy(g->ev_compute);
    hipEventDestroy(g->ev_comm);
    hipStreamDestroy(g->comm_stream);
    delete g;
    return 0;
}

int chase_hip_grid_group_active(chase_hip_grid* g, int group)
{
    if (!g || (group != CHASE_HIP_ROW && group != CHASE_HIP_COL)) return 0;
    return g->active(group) ? 1 : 0;
}

int chase_hip_grid_info(chase_hip_grid* g, int* nprow, int* npcol, int* myrow, int* mycol)
{
    if (!g) return set_error(CHASE_HIP_EINVAL, "grid_info: NULL grid");
    if (nprow) *nprow = g->nprow;
    if (npcol) *npcol = g->npcol;
    if (myrow) *myrow = g->myrow;
    if (mycol) *mycol = g->mycol;
    return 0;
}

// mode 0 = allreduce, 1 = bcast
static int collective(chase_hip_grid* g, int mode, int group, void* dev, size_t count, int root, int async)
{
    if (!g) return set_error(CHASE_HIP_EINVAL, "collective: NULL grid");
    if (group != CHASE_HIP_ROW && group != CHASE_HIP_COL) return set_error(CHASE_HIP_EINVAL, "collective: bad group");
    if (count == 0 || !g->active(group)) return 0;
    if (mode == 1 && (root < 0 || root >= g->group_size(group))) return set_error(CHASE_HIP_EINVAL, "bcast: bad root");
    chase_hip_ctx* c = g->ctx;
    if (g->use_rccl) {
        // comm stream picks up after everything enqueued so far on the compute stream
        HIPCHK(hipEventRecord(g->ev_compute, c->stream));
        HIPCHK(hipStreamWaitEvent(g->comm_stream, g->ev_compute, 0));
        if (mode == 0) NCCLCHK(ncclAllReduce(dev, dev, count, ncclDouble, ncclSum, g->comm[group], g->comm_stream));
        else NCCLCHK(ncclBroadcast(dev, dev, count, ncclDouble, root, g->comm[group], g->comm_stream));
        if (!async) return chase_hip_grid_wait(g);
        return 0;
    }
    // host-callback transport: synchronous by construction
    const size_t bytes = count * sizeof(double);
    int rc = c->ensure_hstage(bytes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->hstage, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = (mode == 0) ? g->h_allreduce(g->h_user, group, (double*)c->hstage, count)
                     : g->h_bcast(g->h_user, group, (double*)c->hstage, count, root);
    if (rc) return set_error(CHASE_HIP_ECOMM, "host transport callback failed");
    HIPCHK(hipMemcpyAsync(dev, c->hstage, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int chase_hip_grid_allreduce(chase_hip_grid* g, int group, void* dev, size_t count, int async)
{
    return collective(g, 0, group, dev, count, 0, async);
}
int chase_hip_grid_bcast(chase_hip_grid* g, int group, void* dev, size_t count, int root, int async)
{
    return collective(g, 1, group, dev, count, root, async);
}
int chase_hip_grid_wait(chase_hip_grid* g)
{
    if (!g) return set_error(CHASE_HIP_EINVAL, "grid_wait: NULL grid");
    if (!g->use_rccl) return 0;
    HIPCHK(hipEventRecord(g->ev_comm, g->comm_stream));
    HIPCHK(hipStreamWaitEvent(g->ctx->stream, g->ev_comm, 0));
    return 0;
}

/* slot events: record = "everything issued so far on the communication stream"; wait = the context (compute) stream waits
 * for the last record of that slot (no-op if the slot was never recorded or the transport is synchronous) */
int chase_hip_grid_event_record(chase_hip_grid* g, int slot)
{
    if (!g || slot < 0) return set_error(CHASE_HIP_EINVAL, "event_record: bad argument");
    if (!g->use_rccl) return 0;
    while ((int)g->slots.size() <= slot) g->slots.push_back(nullptr);
    if (!g->slots[slot]) HIPCHK(hipEventCreateWithFlags(&g->slots[slot], hipEventDisableTiming));
    HIPCHK(hipEventRecord(g->slots[slot], g->comm_stream));
    return 0;
}
int chase_hip_grid_event_wait(chase_hip_grid* g, int slot)
{
    if (!g || slot < 0) return set_error(CHASE_HIP_EINVAL, "event_wait: bad argument");
    if (!g->use_rccl || slot >= (int)g->slots.size() || !g->slots[slot]) return 0;
    HIPCHK(hipStreamWaitEvent(g->ctx->stream, g->slots[slot], 0));
    return 0;
}

int chase_hip_grid_agree_max(chase_hip_grid* g, int* value)
{
    if (!g || !value) return set_error(CHASE_HIP_EINVAL, "agree_max: NULL argument");
    if (g->nprow * g->npcol == 1 && !g->force) return 0;
    // max over all ranks = max over rows of (max over columns); implemented with SUM all-reduces of one-hot-free
    // encoding is not possible, so use two passes of allreduce on (value) via the identity max(a,b) for
    // non-negative ints: we all-reduce the SUM of indicator(value > 0) and of value; control flow only needs
    // "did anybody fail", and the failing info itself for reporting.
    double v[2] = {(double)(*value != 0 ? 1 : 0), (double)*value};
    chase_hip_ctx* c = g->ctx;
    HIPCHK(hipMemcpyAsync(g->scal_dev, v, sizeof v, hipMemcpyHostToDevice, c->stream));
    int rc = chase_hip_grid_allreduce(g, CHASE_HIP_ROW, g->scal_dev, 2, 0);
    if (rc) return rc;
    rc = chase_hip_grid_allreduce(g, CHASE_HIP_COL, g->scal_dev, 2, 0);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(v, g->scal_dev, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (v[0] > 0.5 && *value == 0) *value = (int)(v[1] / v[0] + 0.5) > 0 ? (int)(v[1] / v[0] + 0.5) : 1;
    return 0;
}

/* ---- layout helpers ---------------------------------------------------------------------------------------------- */
long chase_hip_block_len(long n, int p)
{
    if (p <= 0) return n;
    if (n % p == 0) return n / p;
    const long l = n / p + 1;
    return l < n ? l : n;
}
long chase_hip_numroc(long n, long nb, int iproc, int nprocs)
{
    // ScaLAPACK NUMROC with isrcproc = 0
    const long nblocks = n / nb;
    long num = (nblocks / nprocs) * nb;
    const long extra = nblocks % nprocs;
    if (iproc < extra) num += nb;
    else if (iproc == extra) num += n % nb;
    return num;
}
int chase_hip_owner(long g, long nb, int nprocs) { return (int)((g / nb) % nprocs); }
long chase_hip_local_index(long g, long nb, int nprocs) { return (g / (nb * nprocs)) * nb + g % nb; }
long chase_hip_global_index(long l, long nb, int iproc, int nprocs) { return ((l / nb) * nprocs + iproc) * nb + l % nb; }

} // extern "C"
