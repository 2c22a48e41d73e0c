// grid.hip — 2D process grid + collectives (RCCL over xGMI, or host-callback test transport).  See chase_hip_grid.h.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <utility>
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

// Stand-in for a collective on the loopback transport: a few workgroups (RCCL runs its rings on a handful of CUs) make one read
// + write pass over the payload (touch: the HBM traffic a ring all-reduce causes on this device) and then stay resident until
// `ticks` of the constant 100 MHz clock have passed since they started - the time the modelled collective would hold its
// communication stream and its CUs (chase_hip_grid_set_loopback_model).
static int fabric_collective(chase_hip_grid* g, int mode, int group, void* dev, size_t count, int root);
static int fabric_sendrecv(chase_hip_grid* g, int group, const void* sendbuf, size_t sendcount, int peer_send, void* recvbuf,
                           size_t recvcount, int peer_recv);

// FOOTPRINT: an RCCL kernel is 512 threads per workgroup with up to 128 VGPRs per thread (its launch bounds) - eight such
// waves do not fit on a CU beside the TWO workgroups of the MFMA GEMM (2 waves x ~250 VGPRs per SIMD), so a resident collective
// workgroup costs the GEMM a workgroup slot.  The stand-in claims the same: 512 threads, v127 touched.
__global__ __launch_bounds__(512) void loopback_model_kernel(double* x, size_t n, int touch, unsigned long long ticks)
{
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const unsigned long long t0 = (unsigned long long)wall_clock64();
    if (touch) {
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const double v = __builtin_nontemporal_load(x + i);
            __builtin_nontemporal_store(v, x + i);
        }
    }
    while ((unsigned long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
static int loopback_model_launch(hipStream_t st, double* x, size_t n, bool touch, double seconds, double clock_hz, int wgs)
{
    if (!n || (!touch && seconds <= 0)) return 0;
    const int blocks = (int)std::min<size_t>((size_t)std::max(wgs, 1), (n + 511) / 512);
    const unsigned long long ticks = seconds > 0 ? (unsigned long long)(seconds * clock_hz) : 0ull;
    hipLaunchKernelGGL(loopback_model_kernel, dim3(blocks), dim3(512), 0, st, x, n, touch ? 1 : 0, ticks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "loopback_model_kernel");
    return 0;
}

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
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipStreamCreateWithPriority(&g->comm_stream[i], hipStreamNonBlocking, prio_hi));
        HIPCHK(hipEventCreateWithFlags(&g->ev_comm[i], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&g->ev_compute, hipEventDisableTiming));
    if (const char* e = getenv("CHASE_HIP_COMM_STREAMS")) g->nstreams = atoi(e) == 2 ? 2 : 1;
    HIPCHK(hipMalloc((void**)&g->scal_dev, 256));
    return 0;
}

int chase_hip_grid::wait_on(hipEvent_t e)
{
    hipStream_t cs = ctx->stream;
    if (!profiling) {
        HIPCHK(hipStreamWaitEvent(cs, e, 0));
        return 0;
    }
    hipEvent_t a = nullptr, b = nullptr;
    for (hipEvent_t* p : {&a, &b}) {
        if (!ev_pool.empty()) { *p = ev_pool.back(); ev_pool.pop_back(); }
        else if (hipError_t e = hipEventCreate(p); e != hipSuccess) {
            if (a) ev_pool.push_back(a);                 // the first event goes back to the pool, not lost
            return hip_fail(e, "hipEventCreate (exposed-communication bracket)");
        }
    }
    HIPCHK(hipEventRecord(a, cs));
    HIPCHK(hipStreamWaitEvent(cs, e, 0));
    HIPCHK(hipEventRecord(b, cs));
    ev_pending.emplace_back(a, b);
    ++waits;
    if (ev_pending.size() >= 8192) return collect_exposed();
    return 0;
}

int chase_hip_grid::collect_exposed()
{
    if (ev_pending.empty()) return 0;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (auto& pr : ev_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess && ms > 0) exposed_ms += ms;
        ev_pool.push_back(pr.first);
        ev_pool.push_back(pr.second);
    }
    ev_pending.clear();
    return 0;
}

int chase_hip_grid::rccl_failed()
{
    if (!failed.load(std::memory_order_acquire)) return 0;
    return set_error(CHASE_HIP_ECOMM, fail_text[0] ? fail_text : "RCCL transport failed");
}
static long long now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int chase_hip_grid::rccl_issued(int si)
{
    HIPCHK(hipEventRecord(wd_ev[si], comm_stream[si]));
    wd_issue_ns[si].store(now_ns(), std::memory_order_release);
    return 0;
}
// the watchdog of an RCCL grid (see grid.h)
static void rccl_watchdog(chase_hip_grid* g)
{
    (void)hipSetDevice(g->ctx->device);
    auto fail = [&](const char* what) {
        snprintf(g->fail_text, sizeof g->fail_text, "RCCL transport: %s (rank %d of the %d x %d grid): communicators aborted", what,
                 g->rank, g->nprow, g->npcol);
        g->failed.store(1, std::memory_order_release);
        // an enqueue call in flight on the main thread gets two seconds to return; then abort regardless (a call blocked on a
        // dead peer never returns by itself)
        std::unique_lock<std::mutex> lk(g->nccl_mu, std::defer_lock);
        for (int i = 0; i < 200 && !lk.try_lock(); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        for (int i = 0; i < 2; ++i)
            if (g->comm[i]) { (void)ncclCommAbort(g->comm[i]); g->comm[i] = nullptr; }
        fprintf(stderr, "chase_hip: %s\n", g->fail_text);
    };
    while (!g->wd_stop.load(std::memory_order_acquire)) {
        for (int i = 0; i < 2 && !g->failed.load(); ++i) {
            ncclComm_t c = g->comm[i];
            if (!c) continue;
            ncclResult_t ar = ncclSuccess;
            if (ncclCommGetAsyncError(c, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                char b[160];
                snprintf(b, sizeof b, "asynchronous error on the %s communicator: %s", i == CHASE_HIP_ROW ? "row" : "column",
                         ncclGetErrorString(ar));
                fail(b);
            }
        }
        for (int si = 0; si < 2 && !g->failed.load(); ++si) {
            const long long t0 = g->wd_issue_ns[si].load(std::memory_order_acquire);
            if (t0 == 0 || !g->wd_ev[si]) continue;
            if (hipEventQuery(g->wd_ev[si]) != hipErrorNotReady) continue;
            if ((now_ns() - t0) * 1e-9 > g->timeout_s) {
                char b[160];
                snprintf(b, sizeof b, "a collective has not finished %.0f s after it was issued (CHASE_HIP_FABRIC_TIMEOUT_S)", g->timeout_s);
                fail(b);
            }
        }
        if (g->failed.load()) return;
        for (int i = 0; i < 10 && !g->wd_stop.load(std::memory_order_acquire); ++i)
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
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
    if (rc) { chase_hip_grid_destroy(g); return rc; }
    g->use_rccl = true;
    g->force = getenv("CHASE_HIP_RCCL_FORCE") != nullptr;
    // the reference creates its row and column NCCL communicators the same way: one unique id per sub-communicator,
    // ncclCommInitRank on each (grid/mpiGrid2D.hpp:448-484)
    // every failure below releases what was created so far (partially initialised communicators are aborted)
    rc = [&]() -> int {
        if (npcol > 1 || g->force) {
            if (!id_row) return set_error(CHASE_HIP_EINVAL, "grid_create: row id missing");
            ncclUniqueId u; memcpy(&u, id_row, sizeof u);
            NCCLCHK(ncclCommInitRank(&g->comm[CHASE_HIP_ROW], npcol, u, g->mycol));
        }
        if (nprow > 1 || g->force) {
            if (!id_col) return set_error(CHASE_HIP_EINVAL, "grid_create: col id missing");
            ncclUniqueId u; memcpy(&u, id_col, sizeof u);
            NCCLCHK(ncclCommInitRank(&g->comm[CHASE_HIP_COL], nprow, u, g->myrow));
        }
        // first collective on a communicator sets up the xGMI connections (hundreds of ms): pay it here, not in the first
        // filter step, and surface transport problems at construction
        HIPCHK(hipMemsetAsync(g->scal_dev, 0, 64, g->comm_stream[0]));
        HIPCHK(hipStreamSynchronize(g->comm_stream[0]));
        // (each communicator on the stream its collectives will use)
        for (int grp = 0; grp < 2; ++grp)
            if (g->comm[grp]) {
                NCCLCHK(ncclAllReduce(g->scal_dev, g->scal_dev, 8, ncclDouble, ncclSum, g->comm[grp], g->stream_of(grp)));
                HIPCHK(hipStreamSynchronize(g->stream_of(grp)));
            }
        return 0;
    }();
    if (rc) {
        for (int i = 0; i < 2; ++i)
            if (g->comm[i]) { ncclCommAbort(g->comm[i]); g->comm[i] = nullptr; }
        chase_hip_grid_destroy(g);
        return rc;
    }
    if (g->comm[0] || g->comm[1]) {
        if (const char* e = getenv("CHASE_HIP_FABRIC_TIMEOUT_S")) { const double t = atof(e); if (t > 0) g->timeout_s = t; }
        for (int i = 0; i < 2; ++i) {
            g->wd_issue_ns[i].store(0);
            if (hipEventCreateWithFlags(&g->wd_ev[i], hipEventDisableTiming) != hipSuccess) g->wd_ev[i] = nullptr;
        }
        static const bool off = [] { const char* e = getenv("CHASE_HIP_RCCL_WATCHDOG"); return e && atoi(e) == 0; }();
        if (!off && g->wd_ev[0] && g->wd_ev[1]) g->watchdog = std::thread(rccl_watchdog, g);
    }
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
    if (rc) { chase_hip_grid_destroy(g); return rc; }
    g->use_rccl = false;
    g->h_allreduce = allreduce; g->h_bcast = bcast; g->h_user = user;
    *out = g;
    return 0;
}

/* ONE rank of an nprow x npcol grid with nobody on the other side (see grid.h): every collective keeps the stream ordering,
 * events and waits of the RCCL transport and moves nothing. */
int chase_hip_grid_create_loopback(chase_hip_grid** out, chase_hip_ctx* ctx, int nprow, int npcol, int rank)
{
    if (!out) return set_error(CHASE_HIP_EINVAL, "grid_create: NULL out");
    chase_hip_grid* g = new chase_hip_grid();
    int rc = grid_common(g, ctx, nprow, npcol, rank);
    if (rc) { chase_hip_grid_destroy(g); return rc; }
    g->loopback = true;
    {   // rate of wall_clock64 on this device (100 MHz on gfx950; asked, not assumed)
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) == hipSuccess && khz > 0) g->wall_clock_hz = khz * 1e3;
    }
    if (const char* e = getenv("CHASE_HIP_LOOPBACK_TOUCH")) g->loopback_touch = atoi(e) != 0;
    if (const char* e = getenv("CHASE_HIP_LOOPBACK_BUSBW_GBPS")) g->lb_busbw_GBps = atof(e);
    if (const char* e = getenv("CHASE_HIP_LOOPBACK_LATENCY_US")) g->lb_latency_us = atof(e);
    if (const char* e = getenv("CHASE_HIP_LOOPBACK_WGS")) g->lb_wgs = std::max(1, atoi(e));
    *out = g;
    return 0;
}
/* Duration model of the collectives a loopback grid does not perform: every all-reduce / broadcast then holds its communication
 * stream - and `workgroups` workgroups of an RCCL kernel's footprint (512 threads, 128 VGPRs; 0 keeps the current count,
 * default 32) - for latency_us + wire bytes / busbw_GBps (bus bandwidth as RCCL's tests
 * define it: an all-reduce of S bytes among p ranks puts 2 (p - 1) / p S on the wire per rank, a broadcast S); touch != 0 adds
 * one read + write pass over the payload.  busbw_GBps = 0: nothing is enqueued (the compute side alone).  What it is for: the
 * replayed rank's overlap machinery (per-panel events, two streams, exposed-wait brackets) then runs against collectives of
 * a chosen, STATED speed - a model, never a measurement of xGMI. */
int chase_hip_grid_set_loopback_model(chase_hip_grid* g, double busbw_GBps, double latency_us, int touch, int workgroups)
{
    if (!g || !g->loopback) return set_error(CHASE_HIP_EINVAL, "set_loopback_model: not a loopback grid");
    if (busbw_GBps < 0 || latency_us < 0 || workgroups < 0 || workgroups > 1024)
        return set_error(CHASE_HIP_EINVAL, "set_loopback_model: bad argument");
    g->lb_busbw_GBps = busbw_GBps; g->lb_latency_us = latency_us; g->loopback_touch = touch != 0;
    if (workgroups > 0) g->lb_wgs = workgroups;
    return 0;
}

int chase_hip_grid_destroy(chase_hip_grid* g)
{
    if (!g) return 0;
    if (g->ctx) hipSetDevice(g->ctx->device);
    g->wd_stop.store(true, std::memory_order_release);
    if (g->watchdog.joinable()) g->watchdog.join();
    for (int i = 0; i < 2; ++i)
        if (g->comm_stream[i]) (void)hipStreamSynchronize(g->comm_stream[i]);
    for (int i = 0; i < 2; ++i)
        if (g->comm[i]) (void)ncclCommDestroy(g->comm[i]);       // (aborted communicators are nullptr already)
    for (int i = 0; i < 2; ++i)
        if (g->wd_ev[i]) (void)hipEventDestroy(g->wd_ev[i]);
    if (g->scal_dev) (void)hipFree(g->scal_dev);
    if (g->fab_ready) (void)hipEventDestroy(g->fab_ready);
    if (g->fab_done) (void)hipEventDestroy(g->fab_done);
    for (int i = 0; i < 2; ++i)
        for (hipEvent_t e : g->slots[i])
            if (e) (void)hipEventDestroy(e);
    for (auto& pr : g->ev_pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t e : g->ev_pool) (void)hipEventDestroy(e);
    if (g->ev_compute) (void)hipEventDestroy(g->ev_compute);
    for (int i = 0; i < 2; ++i) {
        if (g->ev_comm[i]) (void)hipEventDestroy(g->ev_comm[i]);
        if (g->comm_stream[i]) (void)hipStreamDestroy(g->comm_stream[i]);
    }
    delete g;
    return 0;
}

/* 1 (default): both groups' collectives on one communication stream; 2: one stream per group.  Only between collectives:
 * everything issued so far is waited for first. */
int chase_hip_grid_set_comm_streams(chase_hip_grid* g, int n)
{
    if (!g || (n != 1 && n != 2)) return set_error(CHASE_HIP_EINVAL, "set_comm_streams: 1 or 2");
    if (n == g->nstreams) return 0;
    // everything in flight must have landed before the mapping group -> stream changes (slot events included)
    for (int i = 0; i < 2; ++i) HIPCHK(hipStreamSynchronize(g->comm_stream[i]));
    g->pending[0] = g->pending[1] = false;
    g->nstreams = n;
    return 0;
}
int chase_hip_grid_comm_streams(chase_hip_grid* g) { return g ? g->nstreams : 0; }

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
    if (int f = g->rccl_failed()) return f;
    chase_hip_ctx* c = g->ctx;
    if (c->oplog_on) c->oplog_add(mode == 0 ? "allreduce" : "bcast", group, (long)count, mode == 0 ? 0 : root, async);
    if (g->async_transport()) {
        // this group's comm stream picks up after everything enqueued so far on the compute stream
        hipStream_t cs = g->stream_of(group);
        HIPCHK(hipEventRecord(g->ev_compute, c->stream));
        HIPCHK(hipStreamWaitEvent(cs, g->ev_compute, 0));
        if (g->use_rccl) {
            std::lock_guard<std::mutex> lk(g->nccl_mu);
            if (int f = g->rccl_failed()) return f;
            if (mode == 0) NCCLCHK(ncclAllReduce(dev, dev, count, ncclDouble, ncclSum, g->comm[group], cs));
            else NCCLCHK(ncclBroadcast(dev, dev, count, ncclDouble, root, g->comm[group], cs));
            if (int f = g->rccl_issued(g->stream_index(group))) return f;
        } else if (g->loopback_touch || g->lb_busbw_GBps > 0) {
            const int p = g->group_size(group);
            const double wire = (mode == 0 ? 2.0 * (p - 1) / p : 1.0) * (double)count * sizeof(double);
            const double secs = g->lb_busbw_GBps > 0 ? g->lb_latency_us * 1e-6 + wire / (g->lb_busbw_GBps * 1e9) : 0.0;
            int rc = loopback_model_launch(cs, (double*)dev, count, g->loopback_touch, secs, g->wall_clock_hz, g->lb_wgs);
            if (rc) return rc;
        }
        g->pending[g->stream_index(group)] = true;
        if (!async) return chase_hip_grid_wait(g);
        return 0;
    }
    if (g->fabric) return fabric_collective(g, mode, group, dev, count, root);
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
    if (!g->async_transport()) return 0;
    if (int f = g->rccl_failed()) return f;
    for (int i = 0; i < 2; ++i) {
        if (!g->pending[i]) continue;                   // nothing issued on that stream since the last wait
        HIPCHK(hipEventRecord(g->ev_comm[i], g->comm_stream[i]));
        int rc = g->wait_on(g->ev_comm[i]);
        if (rc) return rc;
        g->pending[i] = false;
    }
    return 0;
}

/* slot events: record = "everything issued so far on the communication stream"; wait = the context (compute) stream waits
 * for the last record of that slot (no-op if the slot was never recorded or the transport is synchronous) */
static int slot_record(chase_hip_grid* g, int si, int slot)
{
    auto& v = g->slots[si];
    while ((int)v.size() <= slot) v.push_back(nullptr);
    if (!v[slot]) HIPCHK(hipEventCreateWithFlags(&v[slot], hipEventDisableTiming));
    HIPCHK(hipEventRecord(v[slot], g->comm_stream[si]));
    return 0;
}
/* record on the communication stream of `group` (the collectives of that group issued so far) */
int chase_hip_grid_event_record_on(chase_hip_grid* g, int group, int slot)
{
    if (!g || slot < 0 || (group != CHASE_HIP_ROW && group != CHASE_HIP_COL))
        return set_error(CHASE_HIP_EINVAL, "event_record: bad argument");
    if (g->ctx->oplog_on) g->ctx->oplog_add("event_record", group, slot, 0, 0);
    if (!g->async_transport()) return 0;
    return slot_record(g, g->stream_index(group), slot);
}
/* record on every communication stream */
int chase_hip_grid_event_record(chase_hip_grid* g, int slot)
{
    if (!g || slot < 0) return set_error(CHASE_HIP_EINVAL, "event_record: bad argument");
    if (!g->async_transport()) return 0;
    for (int si = 0; si < g->nstreams; ++si) { int rc = slot_record(g, si, slot); if (rc) return rc; }
    return 0;
}
/* the context stream waits for the last record of that slot on EVERY communication stream */
int chase_hip_grid_event_wait(chase_hip_grid* g, int slot)
{
    if (!g || slot < 0) return set_error(CHASE_HIP_EINVAL, "event_wait: bad argument");
    if (g->ctx->oplog_on) g->ctx->oplog_add("event_wait", slot, 0, 0, 0);
    if (!g->async_transport()) return 0;
    if (int f = g->rccl_failed()) return f;
    for (int si = 0; si < 2; ++si) {
        if (slot >= (int)g->slots[si].size() || !g->slots[si][slot]) continue;
        int rc = g->wait_on(g->slots[si][slot]);
        if (rc) return rc;
    }
    return 0;
}

int chase_hip_grid_agree_max(chase_hip_grid* g, int* value)
{
    if (!g || !value) return set_error(CHASE_HIP_EINVAL, "agree_max: NULL argument");
    if (g->nprow * g->npcol == 1 && !g->force) return 0;
    if (g->ctx->oplog_on) g->ctx->oplog_add("agree_max", 0, 0, 0, 0);
    if (g->loopback) return 0;                        // nobody to agree with: the replayed rank keeps its own value
    // exact maximum with SUM all-reduces (the one reduction both transports offer): every member of a group writes its
    // value into its own slot of a zeroed vector, the sum is then the list of all values.  Row groups first, column
    // groups on the row maxima second.
    chase_hip_ctx* c = g->ctx;
    struct Mute { chase_hip_ctx* c; Mute(chase_hip_ctx* x) : c(x) { ++c->oplog_mute; } ~Mute() { --c->oplog_mute; } } mute(c);
    int cur = *value;
    for (int grp : {CHASE_HIP_ROW, CHASE_HIP_COL}) {
        if (!g->active(grp)) continue;
        const int sz = g->group_size(grp);
        if (sz > 8) return set_error(CHASE_HIP_EINVAL, "agree_max: group larger than the scratch (8)");
        double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        v[g->group_rank(grp)] = (double)cur;
        HIPCHK(hipMemcpyAsync(g->scal_dev, v, sizeof v, hipMemcpyHostToDevice, c->stream));
        int rc = chase_hip_grid_allreduce(g, grp, g->scal_dev, 8, 0);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(v, g->scal_dev, sizeof v, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (int f = g->rccl_failed()) return f;           // aborted meanwhile: what came back is not a sum
        for (int i = 0; i < sz; ++i) cur = std::max(cur, (int)std::lround(v[i]));
    }
    *value = cur;
    return 0;
}

/* Do all ranks of the grid hold the same 64-bit value (a content hash of something that must be replicated bit for bit)?
 * *all_equal = 1 / 0, identical on every rank.  Same mechanism as agree_max: every member writes (low half, high half,
 * "my row already disagrees") into its own slots of a zeroed vector, SUM all-reduce inside the row groups, then inside the
 * column groups on the row's verdict - two 24-double collectives. */
int chase_hip_grid_agree_equal(chase_hip_grid* g, unsigned long long value, int* all_equal)
{
    if (!g || !all_equal) return set_error(CHASE_HIP_EINVAL, "agree_equal: NULL argument");
    *all_equal = 1;
    if (g->nprow * g->npcol == 1 && !g->force) return 0;
    if (g->ctx->oplog_on) g->ctx->oplog_add("agree_equal", 0, 0, 0, 0);
    if (g->loopback) return 0;
    chase_hip_ctx* c = g->ctx;
    struct Mute { chase_hip_ctx* c; Mute(chase_hip_ctx* x) : c(x) { ++c->oplog_mute; } ~Mute() { --c->oplog_mute; } } mute(c);
    int bad = 0;
    for (int grp : {CHASE_HIP_ROW, CHASE_HIP_COL}) {
        if (!g->active(grp)) continue;
        const int sz = g->group_size(grp), me = g->group_rank(grp);
        if (sz > 8) return set_error(CHASE_HIP_EINVAL, "agree_equal: group larger than the scratch (8)");
        double v[24];
        for (double& x : v) x = 0.0;
        v[me] = (double)(value & 0xffffffffull); v[8 + me] = (double)(value >> 32); v[16 + me] = (double)bad;
        HIPCHK(hipMemcpyAsync(g->scal_dev, v, sizeof v, hipMemcpyHostToDevice, c->stream));
        int rc = chase_hip_grid_allreduce(g, grp, g->scal_dev, 24, 0);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(v, g->scal_dev, sizeof v, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (int f = g->rccl_failed()) return f;
        for (int i = 0; i < sz; ++i)
            if (v[i] != v[0] || v[8 + i] != v[8] || v[16 + i] != 0.0) bad = 1;
    }
    *all_equal = !bad;
    return 0;
}

/* point-to-point exchange inside a group (grid/nccl_utils.hpp:271 ncclSendrecvWrapper): send sendcount doubles to group
 * member peer_send and receive recvcount doubles from peer_recv (a negative peer skips that half; peer == own group rank
 * copies locally).  Ordered after the work enqueued on the context stream; the context stream waits for the result. */
int chase_hip_grid_sendrecv(chase_hip_grid* g, int group, const void* sendbuf, size_t sendcount, int peer_send,
                            void* recvbuf, size_t recvcount, int peer_recv)
{
    if (!g) return set_error(CHASE_HIP_EINVAL, "sendrecv: NULL grid");
    if (group != CHASE_HIP_ROW && group != CHASE_HIP_COL) return set_error(CHASE_HIP_EINVAL, "sendrecv: bad group");
    const int sz = g->group_size(group), me = g->group_rank(group);
    if (peer_send >= sz || peer_recv >= sz) return set_error(CHASE_HIP_EINVAL, "sendrecv: peer outside the group");
    if (sendcount == 0) peer_send = -1;
    if (recvcount == 0) peer_recv = -1;
    chase_hip_ctx* c = g->ctx;
    if (peer_send == me && peer_recv == me) {
        if (sendcount != recvcount) return set_error(CHASE_HIP_EINVAL, "sendrecv: self exchange with different counts");
        HIPCHK(hipMemcpyAsync(recvbuf, sendbuf, sendcount * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    if (peer_send == me || peer_recv == me) return set_error(CHASE_HIP_EINVAL, "sendrecv: half of a self exchange");
    if (peer_send < 0 && peer_recv < 0) return 0;
    if (c->oplog_on) c->oplog_add("sendrecv", group, (long)sendcount, (long)recvcount, 0);
    if (g->loopback) {
        // what would arrive from the peer has the size of what is sent: keep the buffers defined, move nothing elsewhere
        hipStream_t cs = g->stream_of(group);
        HIPCHK(hipEventRecord(g->ev_compute, c->stream));
        HIPCHK(hipStreamWaitEvent(cs, g->ev_compute, 0));
        if (peer_recv >= 0) {
            const size_t n = std::min(sendcount, recvcount);
            if (peer_send >= 0 && n) HIPCHK(hipMemcpyAsync(recvbuf, sendbuf, n * sizeof(double), hipMemcpyDeviceToDevice, cs));
            if (recvcount > n || peer_send < 0)
                HIPCHK(hipMemsetAsync((double*)recvbuf + (peer_send >= 0 ? n : 0), 0,
                                      (recvcount - (peer_send >= 0 ? n : 0)) * sizeof(double), cs));
        }
        g->pending[g->stream_index(group)] = true;
        return chase_hip_grid_wait(g);
    }
    if (g->use_rccl) {
        if (int f = g->rccl_failed()) return f;
        if (!g->comm[group]) return set_error(CHASE_HIP_ECOMM, "sendrecv: group has no communicator");
        hipStream_t cs = g->stream_of(group);
        HIPCHK(hipEventRecord(g->ev_compute, c->stream));
        HIPCHK(hipStreamWaitEvent(cs, g->ev_compute, 0));
        {
            std::lock_guard<std::mutex> lk(g->nccl_mu);
            if (int f = g->rccl_failed()) return f;
            NCCLCHK(ncclGroupStart());
            if (peer_send >= 0) NCCLCHK(ncclSend(sendbuf, sendcount, ncclDouble, peer_send, g->comm[group], cs));
            if (peer_recv >= 0) NCCLCHK(ncclRecv(recvbuf, recvcount, ncclDouble, peer_recv, g->comm[group], cs));
            NCCLCHK(ncclGroupEnd());
            if (int f = g->rccl_issued(g->stream_index(group))) return f;
        }
        g->pending[g->stream_index(group)] = true;
        return chase_hip_grid_wait(g);
    }
    if (g->fabric) return fabric_sendrecv(g, group, sendbuf, sendcount, peer_send, recvbuf, recvcount, peer_recv);
    if (!g->h_sendrecv) return set_error(CHASE_HIP_ECOMM, "sendrecv: host transport has no send/recv callback");
    const size_t sb = (peer_send >= 0 ? sendcount : 0) * sizeof(double), rb = (peer_recv >= 0 ? recvcount : 0) * sizeof(double);
    int rc = c->ensure_hstage(sb + rb + 16);
    if (rc) return rc;
    double* hs = (double*)c->hstage;
    double* hr = hs + (sb / sizeof(double));
    if (sb) HIPCHK(hipMemcpyAsync(hs, sendbuf, sb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (g->h_sendrecv(g->h_user, group, hs, sb / sizeof(double), peer_send, hr, rb / sizeof(double), peer_recv))
        return set_error(CHASE_HIP_ECOMM, "host transport send/recv callback failed");
    if (rb) HIPCHK(hipMemcpyAsync(recvbuf, hr, rb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int chase_hip_grid_set_host_sendrecv(chase_hip_grid* g, chase_hip_host_sendrecv_fn fn)
{
    if (!g) return set_error(CHASE_HIP_EINVAL, "set_host_sendrecv: NULL grid");
    g->h_sendrecv = fn;
    return 0;
}

int chase_hip_grid_set_profiling(chase_hip_grid* g, int on)
{
    if (!g) return set_error(CHASE_HIP_EINVAL, "set_profiling: NULL grid");
    if (!on && g->profiling) { int rc = g->collect_exposed(); if (rc) return rc; }
    g->profiling = on != 0 && g->async_transport();
    return 0;
}
int chase_hip_grid_comm_exposed_ms(chase_hip_grid* g, double* ms, unsigned long long* waits, int reset)
{
    if (!g) return set_error(CHASE_HIP_EINVAL, "comm_exposed_ms: NULL grid");
    int rc = g->collect_exposed();
    if (rc) return rc;
    if (ms) *ms = g->exposed_ms;
    if (waits) *waits = g->waits;
    if (reset) { g->exposed_ms = 0; g->waits = 0; }
    return 0;
}
int chase_hip_grid_transport(chase_hip_grid* g, int* is_rccl, int* row_ranks, int* col_ranks)
{
    if (!g) return set_error(CHASE_HIP_EINVAL, "grid_transport: NULL grid");
    if (is_rccl) *is_rccl = g->use_rccl ? 1 : (g->loopback ? 2 : (g->fabric ? 3 : 0));   /* 0 host callbacks, 1 RCCL, 2 loopback, 3 shared device */
    int n[2] = {1, 1};
    for (int i = 0; i < 2; ++i) {
        if (g->use_rccl && g->comm[i]) NCCLCHK(ncclCommCount(g->comm[i], &n[i]));
        else if (!g->use_rccl) n[i] = g->loopback ? 1 : g->group_size(i);
    }
    if (row_ranks) *row_ranks = n[CHASE_HIP_ROW];
    if (col_ranks) *col_ranks = n[CHASE_HIP_COL];
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

/* ================= shared-device transport: ranks = threads of ONE process on ONE device ==============================
 * (chase_hip_grid.h: chase_hip_fabric_create / chase_hip_grid_create_shared).  A group's all-reduce: every member publishes
 * its buffer and an event on its stream; member 0 sums the buffers in member order into the group's scratch (one kernel,
 * fixed order: every member gets the same bits); every member copies the sum into its buffer on its own stream.  Nothing
 * waits on the host except the threads' rendezvous (condition variables with a time-out; a failed rank aborts the fabric
 * and everybody returns CHASE_HIP_ECOMM instead of waiting). */
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>

namespace {
constexpr int FAB_MAX = 16;
struct FabPtrs { const double* p[FAB_MAX]; int n; };
__global__ __launch_bounds__(256) void fabric_sum_kernel(FabPtrs src, double* __restrict__ out, size_t count)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        double s = src.p[0][i];
        for (int k = 1; k < src.n; ++k) s += src.p[k][i];
        out[i] = s;
    }
}
struct FabBox {                       // one direction of a pair inside a group (send/recv)
    const void* ptr = nullptr; size_t count = 0;
    unsigned long posted = 0, taken = 0;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
};
struct FabGroup {
    int size = 1;
    int arrived = 0;
    unsigned long gen = 0;
    std::vector<void*> ptr;
    std::vector<hipEvent_t> ev_ready, ev_done;        // owned by the members' grids
    void* scratch = nullptr; size_t scratch_bytes = 0;
    hipEvent_t ev_sum = nullptr;
    std::map<std::pair<int, int>, FabBox> box;        // (from, to)
};
} // namespace

struct chase_hip_fabric {
    int nprow = 1, npcol = 1, device = -1;
    std::mutex mu;
    std::condition_variable cv;
    bool failed = false;
    double timeout_s = 600.0;
    std::vector<FabGroup> row, col;                   // row[myrow] has npcol members, col[mycol] has nprow members
    FabGroup& group(int g, int myrow, int mycol) { return g == CHASE_HIP_ROW ? row[myrow] : col[mycol]; }
    // rendezvous of a group's members; returns false on abort / time-out
    bool barrier(FabGroup& G)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const unsigned long my_gen = G.gen;
        if (++G.arrived == G.size) { G.arrived = 0; ++G.gen; cv.notify_all(); return true; }
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return failed || G.gen != my_gen; });
        if (!ok) { failed = true; cv.notify_all(); }
        return ok && !failed;
    }
};

static int fab_fail(const char* what) { return set_error(CHASE_HIP_ECOMM, what); }

static int fabric_collective_body(chase_hip_grid* g, int mode, int group, void* dev, size_t count, int root);
static int fabric_sendrecv_body(chase_hip_grid* g, int group, const void* sendbuf, size_t sendcount, int peer_send, void* recvbuf,
                                size_t recvcount, int peer_recv);
// a member that fails for a reason of its own (a HIP error on its stream, a failed allocation) must not leave the others waiting
// for it until the time-out: whatever fails inside the fabric fails the fabric
static int fab_mark(chase_hip_fabric* F, int rc)
{
    if (rc) { std::lock_guard<std::mutex> lk(F->mu); F->failed = true; F->cv.notify_all(); }
    return rc;
}
static int fabric_collective(chase_hip_grid* g, int mode, int group, void* dev, size_t count, int root)
{
    return fab_mark(g->fabric, fabric_collective_body(g, mode, group, dev, count, root));
}
static int fabric_sendrecv(chase_hip_grid* g, int group, const void* sendbuf, size_t sendcount, int peer_send, void* recvbuf,
                           size_t recvcount, int peer_recv)
{
    return fab_mark(g->fabric, fabric_sendrecv_body(g, group, sendbuf, sendcount, peer_send, recvbuf, recvcount, peer_recv));
}

static int fabric_collective_body(chase_hip_grid* g, int mode, int group, void* dev, size_t count, int root)
{
    chase_hip_fabric* F = g->fabric;
    chase_hip_ctx* c = g->ctx;
    FabGroup& G = F->group(group, g->myrow, g->mycol);
    const int me = g->group_rank(group), p = G.size;
    hipStream_t st = c->stream;
    const size_t bytes = count * sizeof(double);
    // 1. publish my buffer and the point of my stream it is ready at
    HIPCHK(hipEventRecord(g->fab_ready, st));
    { std::lock_guard<std::mutex> lk(F->mu); G.ptr[me] = dev; G.ev_ready[me] = g->fab_ready; G.ev_done[me] = g->fab_done; }
    if (!F->barrier(G)) return fab_fail("shared transport: a rank failed or timed out (publish)");
    if (mode == 0) {
        if (me == 0) {
            // the scratch is free once every member's previous copy out of it has run (their ev_done of the previous round)
            for (int k = 0; k < p; ++k) HIPCHK(hipStreamWaitEvent(st, G.ev_done[k], 0));
            if (G.scratch_bytes < bytes) {
                HIPCHK(hipStreamSynchronize(st));
                if (G.scratch) HIPCHK(hipFree(G.scratch));
                G.scratch = nullptr; G.scratch_bytes = 0;
                HIPCHK(hipMalloc(&G.scratch, bytes));
                G.scratch_bytes = bytes;
            }
            FabPtrs src; src.n = p;
            for (int k = 0; k < p; ++k) { HIPCHK(hipStreamWaitEvent(st, G.ev_ready[k], 0)); src.p[k] = (const double*)G.ptr[k]; }
            const unsigned blocks = (unsigned)std::min<size_t>(2048, (count + 255) / 256);
            hipLaunchKernelGGL(fabric_sum_kernel, dim3(blocks), dim3(256), 0, st, src, (double*)G.scratch, count);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(G.ev_sum, st));
        }
        if (!F->barrier(G)) return fab_fail("shared transport: a rank failed or timed out (sum)");
        HIPCHK(hipStreamWaitEvent(st, G.ev_sum, 0));
        HIPCHK(hipMemcpyAsync(dev, G.scratch, bytes, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipEventRecord(g->fab_done, st));
        if (!F->barrier(G)) return fab_fail("shared transport: a rank failed or timed out (copy)");
        return 0;
    }
    // broadcast: everybody copies the root's buffer; the root must not touch it again before the copies have run
    if (me != root) {
        HIPCHK(hipStreamWaitEvent(st, G.ev_ready[root], 0));
        HIPCHK(hipMemcpyAsync(dev, G.ptr[root], bytes, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(hipEventRecord(g->fab_done, st));
    if (!F->barrier(G)) return fab_fail("shared transport: a rank failed or timed out (broadcast)");
    if (me == root)
        for (int k = 0; k < p; ++k)
            if (k != root) HIPCHK(hipStreamWaitEvent(st, G.ev_done[k], 0));
    // (the members' ev_done now mean "broadcast copied", which also satisfies the next all-reduce's scratch wait)
    return 0;
}

static int fabric_sendrecv_body(chase_hip_grid* g, int group, const void* sendbuf, size_t sendcount, int peer_send, void* recvbuf,
                                size_t recvcount, int peer_recv)
{
    chase_hip_fabric* F = g->fabric;
    FabGroup& G = F->group(group, g->myrow, g->mycol);
    const int me = g->group_rank(group);
    hipStream_t st = g->ctx->stream;
    FabBox* out = nullptr;
    if (peer_send >= 0) {
        std::unique_lock<std::mutex> lk(F->mu);
        out = &G.box[{me, peer_send}];
        if (!out->ev_ready) {
            lk.unlock();
            hipEvent_t a = nullptr, b = nullptr;
            HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            lk.lock();
            out->ev_ready = a; out->ev_done = b;
        }
        hipEvent_t ready = out->ev_ready;
        lk.unlock();
        HIPCHK(hipEventRecord(ready, st));
        lk.lock();
        out->ptr = sendbuf; out->count = sendcount; ++out->posted;
        F->cv.notify_all();
    }
    if (peer_recv >= 0) {
        std::unique_lock<std::mutex> lk(F->mu);
        FabBox& in = G.box[{peer_recv, me}];
        const bool ok = F->cv.wait_for(lk, std::chrono::duration<double>(F->timeout_s), [&] { return F->failed || in.posted > in.taken; });
        if (!ok || F->failed) { F->failed = true; F->cv.notify_all(); return fab_fail("shared transport: peer never sent"); }
        if (in.count != recvcount) { F->failed = true; F->cv.notify_all(); return fab_fail("shared transport: send / receive counts differ"); }
        const void* src = in.ptr; hipEvent_t ready = in.ev_ready, done = in.ev_done;
        lk.unlock();
        HIPCHK(hipStreamWaitEvent(st, ready, 0));
        HIPCHK(hipMemcpyAsync(recvbuf, src, recvcount * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipEventRecord(done, st));
        lk.lock();
        ++in.taken;
        F->cv.notify_all();
    }
    if (out) {
        std::unique_lock<std::mutex> lk(F->mu);
        const bool ok = F->cv.wait_for(lk, std::chrono::duration<double>(F->timeout_s), [&] { return F->failed || out->taken == out->posted; });
        if (!ok || F->failed) { F->failed = true; F->cv.notify_all(); return fab_fail("shared transport: peer never received"); }
        hipEvent_t done = out->ev_done;
        lk.unlock();
        HIPCHK(hipStreamWaitEvent(st, done, 0));               // my send buffer is free again once the peer's copy has run
    }
    return 0;
}

extern "C" {

int chase_hip_fabric_create(chase_hip_fabric** out, int nprow, int npcol)
{
    if (!out || nprow < 1 || npcol < 1 || nprow > FAB_MAX || npcol > FAB_MAX) return set_error(CHASE_HIP_EINVAL, "fabric_create: bad grid");
    chase_hip_fabric* F = new chase_hip_fabric();
    F->nprow = nprow; F->npcol = npcol;
    F->row.resize(nprow); F->col.resize(npcol);
    for (auto& G : F->row) { G.size = npcol; G.ptr.assign(npcol, nullptr); G.ev_ready.assign(npcol, nullptr); G.ev_done.assign(npcol, nullptr); }
    for (auto& G : F->col) { G.size = nprow; G.ptr.assign(nprow, nullptr); G.ev_ready.assign(nprow, nullptr); G.ev_done.assign(nprow, nullptr); }
    if (const char* e = getenv("CHASE_HIP_FABRIC_TIMEOUT_S")) F->timeout_s = atof(e);
    *out = F;
    return 0;
}
/* a rank failed: every thread waiting in the fabric (and every later call) returns CHASE_HIP_ECOMM */
int chase_hip_fabric_abort(chase_hip_fabric* F)
{
    if (!F) return 0;
    std::lock_guard<std::mutex> lk(F->mu);
    F->failed = true;
    F->cv.notify_all();
    return 0;
}
/* after every grid on it is destroyed */
int chase_hip_fabric_destroy(chase_hip_fabric* F)
{
    if (!F) return 0;
    if (F->device >= 0) (void)hipSetDevice(F->device);
    for (auto* v : {&F->row, &F->col})
        for (auto& G : *v) {
            if (G.scratch) (void)hipFree(G.scratch);
            if (G.ev_sum) (void)hipEventDestroy(G.ev_sum);
            for (auto& kv : G.box) { if (kv.second.ev_ready) (void)hipEventDestroy(kv.second.ev_ready); if (kv.second.ev_done) (void)hipEventDestroy(kv.second.ev_done); }
        }
    delete F;
    return 0;
}

int chase_hip_grid_create_shared(chase_hip_grid** out, chase_hip_ctx* ctx, int nprow, int npcol, int rank, chase_hip_fabric* F)
{
    if (!out || !F) return set_error(CHASE_HIP_EINVAL, "grid_create_shared: NULL argument");
    if (F->nprow != nprow || F->npcol != npcol) return set_error(CHASE_HIP_EINVAL, "grid_create_shared: the fabric is for another grid");
    chase_hip_grid* g = new chase_hip_grid();
    int rc = grid_common(g, ctx, nprow, npcol, rank);
    if (rc) { chase_hip_grid_destroy(g); return rc; }
    g->fabric = F;
    rc = [&]() -> int {
        HIPCHK(hipEventCreateWithFlags(&g->fab_ready, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&g->fab_done, hipEventDisableTiming));
        std::lock_guard<std::mutex> lk(F->mu);
        if (F->device >= 0 && F->device != ctx->device) return set_error(CHASE_HIP_EINVAL, "grid_create_shared: the ranks of a fabric share ONE device");
        F->device = ctx->device;
        for (int grp : {CHASE_HIP_ROW, CHASE_HIP_COL}) {
            FabGroup& G = F->group(grp, g->myrow, g->mycol);
            if (!G.ev_sum) HIPCHK(hipEventCreateWithFlags(&G.ev_sum, hipEventDisableTiming));
            const int me = g->group_rank(grp);
            G.ev_ready[me] = g->fab_ready; G.ev_done[me] = g->fab_done;
        }
        return 0;
    }();
    if (rc) { chase_hip_grid_destroy(g); return rc; }
    // every member's events must exist before anybody waits on them: the first collective's publish rendezvous orders that
    *out = g;
    return 0;
}

} // extern "C"
