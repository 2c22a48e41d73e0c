// grid.h — the process grid behind chase_hip_grid* (include/chase_hip_grid.h)
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/chase_hip_grid.h"
#include "ctx.h"

// In-process fabric of rank THREADS that share one device (chase_hip_fabric_create): the collectives of such ranks are device-side
// sums / copies between the ranks' buffers, ordered by events between the ranks' streams - no host staging, no host
// synchronisation.  Test plumbing like the host-callback transport (N ranks on ONE GPU), but asynchronous like RCCL.
struct chase_hip_fabric;

struct chase_hip_grid {
    chase_hip_ctx* ctx = nullptr;
    int nprow = 1, npcol = 1, rank = 0, myrow = 0, mycol = 0;
    bool use_rccl = false;
    chase_hip_fabric* fabric = nullptr;               // shared-device transport (ranks = threads of this process on one GPU)
    hipEvent_t fab_ready = nullptr, fab_done = nullptr;
    // loopback: a transport for ONE rank of a larger grid with nothing on the other side (single-rank replay of a
    // multi-GPU solve, bench.py --replay-rank): collectives keep their stream ordering, events and waits exactly as with
    // RCCL but enqueue no communication (optionally one read+write pass over the payload, loopback_touch: the HBM traffic a
    // ring all-reduce causes on this device).  Results are wrong by construction; the compute side's time is right.
    bool loopback = false;
    bool loopback_touch = false;
    // duration model of the absent collectives (chase_hip_grid_set_loopback_model): a collective occupies the communication
    // stream - and a collective's share of CUs - for latency + wire bytes / bus bandwidth (RCCL's definition of bus bandwidth:
    // an all-reduce of S bytes among p ranks moves 2 (p - 1) / p S per rank); 0 = nothing is enqueued
    double lb_busbw_GBps = 0.0, lb_latency_us = 0.0, wall_clock_hz = 1e8;
    int lb_wgs = 32;                                  // workgroups of the stand-in kernel (RCCL: one per channel)
    bool force = false;                               // CHASE_HIP_RCCL_FORCE: run size-1 groups through RCCL too (testing)
    ncclComm_t comm[2] = {nullptr, nullptr};          // [ROW], [COL]
    // Communication streams: ONE for both groups by default; optionally one per group (CHASE_HIP_COMM_STREAMS=2 /
    // chase_hip_grid_set_comm_streams(2)): on a 4 x 2 grid the row and column communicators use disjoint xGMI links, so
    // their collectives need not queue behind each other.  Why not the default: measured over RCCL's socket transport
    // (profiles/r05_socket_rccl_streams.txt), collectives that alternate between two communicators on two streams cost
    // ~19 ms each instead of ~1 ms on one stream (this RCCL orders the launches of a process's communicators host-side) -
    // nobody has seen what xGMI does, so two streams are a candidate the first-contact self-tuning measures, not a bet.
    hipStream_t comm_stream[2] = {nullptr, nullptr};
    int nstreams = 1;
    bool pending[2] = {false, false};                 // collectives issued on stream i since the compute stream last waited
    hipEvent_t ev_compute = nullptr, ev_comm[2] = {nullptr, nullptr};
    hipStream_t stream_of(int group) const { return comm_stream[nstreams == 2 ? group : 0]; }
    int stream_index(int group) const { return nstreams == 2 ? group : 0; }
    bool async_transport() const { return use_rccl || loopback; }
    chase_hip_host_allreduce_fn h_allreduce = nullptr;
    chase_hip_host_bcast_fn h_bcast = nullptr;
    chase_hip_host_sendrecv_fn h_sendrecv = nullptr;
    void* h_user = nullptr;
    double* scal_dev = nullptr;                        // scratch of the agreement collectives (32 doubles)
    // Failure surface of the RCCL transport (round 6; the reference exits on the first NCCL error, grid/nccl_utils.hpp:13-26).
    // A watchdog thread polls ncclCommGetAsyncError on both communicators and the age of the newest unfinished collective
    // (wd_ev: recorded behind every collective on its communication stream); on an asynchronous error - a dead peer - or after
    // CHASE_HIP_FABRIC_TIMEOUT_S (default 600) without progress it ABORTS both communicators (their kernels then leave the
    // device, whatever the host is blocked in returns) and latches `failed`: every later grid call returns CHASE_HIP_ECOMM with
    // fail_text, which the Impls turn into an exception - the rank process ends non-zero instead of hanging.
    std::thread watchdog;
    std::atomic<bool> wd_stop{false};
    std::atomic<int> failed{0};
    std::mutex nccl_mu;                                // enqueue calls vs. the watchdog's abort
    char fail_text[256] = {0};
    hipEvent_t wd_ev[2] = {nullptr, nullptr};
    std::atomic<long long> wd_issue_ns[2];             // when the newest collective of stream i was enqueued (0: none yet)
    double timeout_s = 600.0;
    int rccl_failed();                                 // != 0: sets the error text and returns CHASE_HIP_ECOMM
    int rccl_issued(int si);                           // bookkeeping behind an enqueued collective
    std::vector<hipEvent_t> slots[2];                  // per-panel 'all-reduce done' events (pipelined HEMM), per stream
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
    chase_hip_ctx* octx() const { return ctx; }
    int group_rank(int g) const { return g == CHASE_HIP_ROW ? mycol : myrow; }
};
