// grid.h — the process grid behind chase_hip_grid* (include/chase_hip_grid.h)
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <vector>
#include "../../include/chase_hip_grid.h"
#include "ctx.h"

struct chase_hip_grid {
    chase_hip_ctx* ctx = nullptr;
    int nprow = 1, npcol = 1, rank = 0, myrow = 0, mycol = 0;
    bool use_rccl = false;
    bool force = false;                               // CHASE_HIP_RCCL_FORCE: run size-1 groups through RCCL too (testing)
    ncclComm_t comm[2] = {nullptr, nullptr};          // [ROW], [COL]
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_compute = nullptr, ev_comm = nullptr;
    chase_hip_host_allreduce_fn h_allreduce = nullptr;
    chase_hip_host_bcast_fn h_bcast = nullptr;
    chase_hip_host_sendrecv_fn h_sendrecv = nullptr;
    void* h_user = nullptr;
    double* scal_dev = nullptr;                        // one double for agree_max
    std::vector<hipEvent_t> slots;                     // per-panel 'all-reduce done' events (pipelined HEMM)
    // profiling of exposed communication: every wait of the compute stream on the communication stream is bracketed by
    // two timing events; their distance is the time the compute stream had nothing to do but wait
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
    double exposed_ms = 0;
    unsigned long long waits = 0;
    int wait_on(hipEvent_t e);                         // compute stream waits for e (bracketed when profiling)
    int collect_exposed();                             // synchronises the compute stream, folds the pending brackets in
    int group_size(int g) const { return g == CHASE_HIP_ROW ? npcol : nprow; }
    bool active(int g) const { return group_size(g) > 1 || (force && use_rccl); }
    int group_rank(int g) const { return g == CHASE_HIP_ROW ? mycol : myrow; }
};
