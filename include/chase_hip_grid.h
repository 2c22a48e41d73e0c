/* chase_hip_grid.h — C ABI of the 2D process grid and its collectives (one process per GPU).
 *
 * Replaces grid/mpiGrid2D.hpp (MpiGrid2D<ColMajor>: Cartesian grid, row/col communicators, three ncclComm_t) and the
 * typed wrappers of grid/nccl_utils.hpp:200-278.  Rank -> coordinates is the reference's column-major grid ordering
 * (rank = row + col * nprow, grid/mpiGrid2D.hpp:402-446).  Two transports:
 *   - RCCL over xGMI (production): ncclAllReduce / ncclBroadcast of doubles on a dedicated HIP stream;
 *   - host callbacks (test plumbing): the payload is staged through pinned host memory and handed to a callback (the tests
 *     supply one that meets the other ranks' callbacks inside the process - ranks as threads - or torch.distributed/gloo -
 *     ranks as processes), which lets N ranks share ONE GPU so that the distributed code path is exercised on a
 *     single-GPU box.  A rank is whatever calls the collectives in the same order as its peers: a process or a thread.
 * Groups: CHASE_HIP_ROW = ranks sharing a grid row (size npcol, the reference's row_comm),
 *         CHASE_HIP_COL = ranks sharing a grid column (size nprow, the reference's col_comm).
 * Data layout helpers implement the reference's block rule (linalg/distMatrix/distMatrix.hpp:1992-2052) and numroc
 * (:44-67); a block layout is the block-cyclic layout with block size = block length. */
#ifndef CHASE_HIP_GRID_H
#define CHASE_HIP_GRID_H
#include <stddef.h>
#include "chase_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

#define CHASE_HIP_ROW 0
#define CHASE_HIP_COL 1
#define CHASE_HIP_UNIQUE_ID_BYTES 128

typedef struct chase_hip_grid chase_hip_grid;
typedef int (*chase_hip_host_allreduce_fn)(void* user, int group, double* buf, size_t count);
typedef int (*chase_hip_host_bcast_fn)(void* user, int group, double* buf, size_t count, int root);
/* send `sendcount` doubles to group member peer_send (< 0: nothing to send) and receive recvcount from peer_recv */
typedef int (*chase_hip_host_sendrecv_fn)(void* user, int group, const double* sendbuf, size_t sendcount, int peer_send,
                                          double* recvbuf, size_t recvcount, int peer_recv);

int chase_hip_rccl_unique_id(char id[CHASE_HIP_UNIQUE_ID_BYTES]);
/* id_row / id_col: the unique ids of THIS rank's row group and column group (ignored when that group has size 1) */
int chase_hip_grid_create_rccl(chase_hip_grid** out, chase_hip_ctx* ctx, int nprow, int npcol, int rank,
                               const char id_row[CHASE_HIP_UNIQUE_ID_BYTES],
                               const char id_col[CHASE_HIP_UNIQUE_ID_BYTES]);
int chase_hip_grid_create_host(chase_hip_grid** out, chase_hip_ctx* ctx, int nprow, int npcol, int rank,
                               chase_hip_host_allreduce_fn allreduce, chase_hip_host_bcast_fn bcast, void* user);
/* ONE rank of an nprow x npcol grid with nobody on the other side: all-reduce / broadcast / send-recv keep the stream
 * ordering, events and waits of the RCCL transport (communication streams, per-panel slots, exposed-wait brackets) and move
 * nothing (CHASE_HIP_LOOPBACK_TOUCH=1: one read + write pass over the payload on the communication stream - the HBM traffic
 * a ring all-reduce causes on this device); agree_max keeps the caller's value.  For the single-rank REPLAY of a multi-GPU
 * solve (bench.py --replay-rank 4x2): results are wrong by construction, the compute side's time is right. */
int chase_hip_grid_create_loopback(chase_hip_grid** out, chase_hip_ctx* ctx, int nprow, int npcol, int rank);
/* Duration model of the collectives a loopback grid does not perform: each all-reduce / broadcast holds its communication
 * stream and `workgroups` workgroups with an RCCL kernel's footprint (512 threads, 128 VGPRs: they cannot share a CU with two
 * workgroups of the MFMA GEMM; 0 = keep, default 32) for latency_us + wire bytes / busbw_GBps (RCCL's bus bandwidth: an
 * all-reduce of S bytes among p ranks puts 2 (p - 1) / p S on the wire per rank, a broadcast S); touch: plus one read + write
 * pass over the payload.  0 GB/s = nothing enqueued.  A stated MODEL for exercising the overlap machinery of the replayed
 * rank, never a measurement of xGMI (env: CHASE_HIP_LOOPBACK_BUSBW_GBPS / _LATENCY_US / _TOUCH / _WGS). */
int chase_hip_grid_set_loopback_model(chase_hip_grid* g, double busbw_GBps, double latency_us, int touch, int workgroups);
/* Shared-device transport (test plumbing, like the host callbacks): the ranks are THREADS of one process that share ONE device;
 * a group's all-reduce is a device-side sum of the members' buffers in member order (every member gets the same bits), a
 * broadcast / send-recv a device-to-device copy, all ordered by events between the ranks' streams - no host staging, no host
 * synchronisation.  One fabric object per grid, created before the rank threads start; a failing rank aborts it, and every
 * rank waiting in a collective returns CHASE_HIP_ECOMM instead of hanging (CHASE_HIP_FABRIC_TIMEOUT_S, default 600). */
typedef struct chase_hip_fabric chase_hip_fabric;
int chase_hip_fabric_create(chase_hip_fabric** out, int nprow, int npcol);
int chase_hip_fabric_abort(chase_hip_fabric* f);
int chase_hip_fabric_destroy(chase_hip_fabric* f);
int chase_hip_grid_create_shared(chase_hip_grid** out, chase_hip_ctx* ctx, int nprow, int npcol, int rank, chase_hip_fabric* f);
int chase_hip_grid_destroy(chase_hip_grid* g);
/* communication streams: 1 (default) = both groups on one stream; 2 = one per group - on a 4 x 2 grid the row and column
 * communicators use disjoint xGMI links, so their collectives need not queue behind each other (CHASE_HIP_COMM_STREAMS=2).
 * Not the default because RCCL's socket transport shows a 20x latency penalty for collectives alternating between two
 * communicators on two streams (profiles/r05_socket_rccl_streams.txt); the multi-GPU bench measures both at first contact.
 * Callable between collectives (it synchronises the communication streams). */
int chase_hip_grid_set_comm_streams(chase_hip_grid* g, int n);
int chase_hip_grid_comm_streams(chase_hip_grid* g);
int chase_hip_grid_info(chase_hip_grid* g, int* nprow, int* npcol, int* myrow, int* mycol);
/* 1 if collectives in `group` really communicate (group size > 1, or CHASE_HIP_RCCL_FORCE set: size-1 groups are then
 * run through RCCL as well so that a single-GPU box exercises ncclCommInitRank / ncclAllReduce / the stream logic) */
int chase_hip_grid_group_active(chase_hip_grid* g, int group);
/* in-place SUM all-reduce / broadcast of `count` doubles of device memory inside `group`; ordered after the work
 * already enqueued on the context stream, and the context stream waits for the result (async = 0), or the caller
 * orders it with chase_hip_grid_wait (async != 0: the collective runs on the grid's communication stream). */
int chase_hip_grid_allreduce(chase_hip_grid* g, int group, void* dev, size_t count, int async);
int chase_hip_grid_bcast(chase_hip_grid* g, int group, void* dev, size_t count, int root, int async);
int chase_hip_grid_wait(chase_hip_grid* g); /* context stream waits for all collectives issued so far */
/* per-slot events for pipelining: record on the communication stream / make the context stream wait for a slot */
int chase_hip_grid_event_record(chase_hip_grid* g, int slot);                /* on every communication stream */
int chase_hip_grid_event_record_on(chase_hip_grid* g, int group, int slot);  /* on the stream of `group`'s collectives */
int chase_hip_grid_event_wait(chase_hip_grid* g, int slot);                  /* last record of the slot on every stream */
/* all ranks agree on the maximum of a host integer (control-flow decisions such as the potrf info) */
int chase_hip_grid_agree_max(chase_hip_grid* g, int* value);
/* all ranks learn whether they all hold the same 64-bit value (e.g. chase_hip_hash64 of a matrix that every rank computed for
 * itself and that must be replicated bit for bit): *all_equal = 1 / 0, the same on every rank */
int chase_hip_grid_agree_equal(chase_hip_grid* g, unsigned long long value, int* all_equal);
/* point-to-point exchange of device doubles inside `group` — replaces ncclSendrecvWrapper / MPI_Sendrecv
 * (grid/nccl_utils.hpp:271, linalg/distMatrix/distMultiVector.hpp:1944-1958): send to group member peer_send, receive
 * from peer_recv (negative peer or zero count: that half is skipped; both peers == own group rank: local copy).  RCCL:
 * ncclSend + ncclRecv in one group call on the communication stream. */
int chase_hip_grid_sendrecv(chase_hip_grid* g, int group, const void* sendbuf, size_t sendcount, int peer_send,
                            void* recvbuf, size_t recvcount, int peer_recv);
int chase_hip_grid_set_host_sendrecv(chase_hip_grid* g, chase_hip_host_sendrecv_fn fn); /* host transport only */
/* exposed communication: with profiling on, every wait of the context (compute) stream on the communication stream is
 * bracketed by timing events; comm_exposed_ms returns the accumulated time the compute stream spent waiting with nothing
 * else to run (synchronises the context stream), and the number of waits */
int chase_hip_grid_set_profiling(chase_hip_grid* g, int on);
int chase_hip_grid_comm_exposed_ms(chase_hip_grid* g, double* ms, unsigned long long* waits, int reset);
/* which transport the grid runs on (*is_rccl: 0 host callbacks, 1 RCCL, 2 loopback, 3 shared device) and how many ranks RCCL itself reports for this rank's row / column communicator
 * (ncclCommCount; 1 for a group without communicator) */
int chase_hip_grid_transport(chase_hip_grid* g, int* is_rccl, int* row_ranks, int* col_ranks);

/* Householder QR of a row-distributed block inside `group`: V_loc (mloc x n, ldv) <- this rank's rows of the first n
 * columns of Q, V = Q R over all the group's rows (their total must be >= n).  row_offset = rows held by the members with
 * a lower group rank (the pivot of column j is row j of that stacked order; block and block-cyclic row layouts alike).
 * Replaces cpu_distributed_houseQR_formQ / houseQR1_formQ (linalg/internal/mpi/householder_qr.hpp:737-1417,
 * nccl/householder_qr.hpp:2957): one fused all-reduce per column, two per panel, scalars on the device, no buffer larger
 * than the local block. */
int chase_hip_houseqr_dist(chase_hip_ctx* ctx, chase_hip_grid* g, int group, int cplx, int mloc, int n, void* V, long ldv,
                           long row_offset);

/* ---- layout helpers (pure host arithmetic, no GPU needed) ------------------------------------------------------ */
long chase_hip_block_len(long n, int nprocs);                      /* distMatrix.hpp:2000-2007 */
long chase_hip_numroc(long n, long nb, int iproc, int nprocs);     /* distMatrix.hpp:44-67 (isrcproc = 0) */
int chase_hip_owner(long g, long nb, int nprocs);                  /* (g / nb) % nprocs */
long chase_hip_local_index(long g, long nb, int nprocs);           /* (g / (nb*nprocs)) * nb + g % nb */
long chase_hip_global_index(long l, long nb, int iproc, int nprocs);

#ifdef __cplusplus
}
#endif
#endif
