/* chase_hip_solver.h — C ABI of the host solver built on include/chase_hip.h.
 *
 * One opaque handle wraps a ChaseHip<T> Impl (chase_amd/host/chase_hip_impl.hpp), i.e. an implementation of the
 * reference's ChaseBase<T> operator surface (algorithm/interface.hpp:46-434), plus the driver (chase::Solve,
 * algorithm/algorithm.hpp:345-364).  The chase_hip_op_* entry points expose the virtuals one by one with the
 * reference's argument meaning; chase_hip_solver_solve runs the whole ChASE iteration.
 * Same status convention as chase_hip.h. */
#ifndef CHASE_HIP_SOLVER_H
#define CHASE_HIP_SOLVER_H
#include <stddef.h>
#include "chase_hip.h"
#include "chase_hip_grid.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct chase_hip_solver chase_hip_solver;

typedef struct chase_hip_stats {
    size_t iterations, filtered_vecs, lanczos_vecs, locked;
    double t_all, t_init, t_lanczos, t_filter, t_qr, t_rr, t_resid; /* host wall seconds per phase */
    double filter_ms_device;                                         /* HIP-event time between FilterPhaseStart/End */
    double lowerb, upperb, lambda;
} chase_hip_stats;

/* H (N x N, ldh), V (N x (nev+nex), ldv) column-major; ritzv: nev+nex doubles.  cplx: 0 = double, 1 = complex double.
 * H and V are host pointers owned by the caller (ChASECPU/ChASEGPU constructor contract, chase_cpu.hpp:73-74);
 * h_on_device != 0 declares H a device pointer that is used in place. */
int chase_hip_solver_create(chase_hip_solver** out, chase_hip_ctx* ctx, int cplx, size_t N, size_t nev, size_t nex,
                            void* H, size_t ldh, void* V, size_t ldv, double* ritzv, int h_on_device);
/* Pseudo-Hermitian (Bethe-Salpeter) sequential Impl — ChASECPU<T, PseudoHermitianMatrix<T>> contract
 * (Impl/chase_cpu/chase_cpu.hpp:78-97): V is N x 2*(nev+nex), ritzv holds 2*(nev+nex) values; chase_hip_solver_solve then
 * runs chase::Solve_pseudo (algorithm/algorithm.inc:1834-2220). */
int chase_hip_solver_create_pseudo(chase_hip_solver** out, chase_hip_ctx* ctx, int cplx, size_t N, size_t nev, size_t nex,
                                   void* H, size_t ldh, void* V, size_t ldv, double* ritzv, int h_on_device);
/* Distributed Impl (pChASECPU / pChASEGPU constructor contract, pchase_cpu.hpp:92-93): H_loc is this rank's DEVICE
 * block of the N x N matrix, distributed block-cyclically with (mb, nb) over the grid (mb = nb = 0: the reference's block
 * layout); the nev+nex vectors live distributed on the devices (chase_hip_psolver_{upload,download}_v move a rank's
 * m_loc x (nev+nex) block).  All op / solve entry points below are collective over the grid. */
int chase_hip_psolver_create(chase_hip_solver** out, chase_hip_ctx* ctx, chase_hip_grid* grid, int cplx, size_t N,
                             size_t nev, size_t nex, size_t mb, size_t nb, void* H_loc_dev, size_t ldh, double* ritzv);
/* distributed pseudo-Hermitian (Bethe-Salpeter) Impl — mirrors pChASECPU / pChASEGPU over
 * PseudoHermitianBlockBlockMatrix / PseudoHermitianBlockCyclicMatrix (Impl/pchase_cpu/pchase_cpu.hpp:92-190):
 * the vector blocks have 2*(nev+nex) columns and ritzv 2*(nev+nex) entries; N must be even */
int chase_hip_psolver_create_pseudo(chase_hip_solver** out, chase_hip_ctx* ctx, chase_hip_grid* grid, int cplx, size_t N,
                                    size_t nev, size_t nex, size_t mb, size_t nb, void* H_loc_dev, size_t ldh,
                                    double* ritzv);
int chase_hip_psolver_local_shape(chase_hip_solver* s, size_t* m_loc, size_t* n_loc);
int chase_hip_psolver_upload_v(chase_hip_solver* s, const void* host, size_t ldv);
int chase_hip_psolver_download_v(chase_hip_solver* s, void* host, size_t ldv);
int chase_hip_psolver_set_pipeline(chase_hip_solver* s, int on); /* 0: no compute/communication overlap (debug) */
int chase_hip_solver_destroy(chase_hip_solver* s);
/* keys: tol deg maxdeg degextra maxiter lanczositer numlanczos opt approx cholqr decayingrate clusteraware upperbscale
 * (ChaseConfig setters, algorithm/configuration.hpp:197-462); get additionally: locked qr_variant filter_ms hemm_calls
 * hemm_reused_vecs resd_rechecked, and iterations / filtered_vecs of the last solve.
 * Grid Impls, run-time knobs of the panel-pipelined HEMM (set and get; COLLECTIVE: the same value on every rank at the same
 * point): panel_cols (a multiple of 64 in [64, 4096]; default: the width whose GEMM fills the chip once with whole tiles, computed
 * from the layout's largest local block, but at most 1 / 2.5 of nev + nex and not below 128 - a product must consist of several
 * panels for any of its all-reduce to hide), panel_rounds (K pieces of a panel product that shares the chip with a collective,
 * 0..16, default 4 = CHASE_HIP_PANEL_ROUNDS), pipeline (0: every all-reduce waited for where it is issued). */
int chase_hip_solver_set(chase_hip_solver* s, const char* key, double value);
int chase_hip_solver_get(chase_hip_solver* s, const char* key, double* value);
int chase_hip_solver_solve(chase_hip_solver* s, int record_trace);
/* Observer of the outer iterations of chase_hip_solver_solve (the while loop of algorithm/algorithm.inc:1491-1720): called
 * after Lock() of every iteration with the 0-based iteration index, the vectors filtered in it, and the locked / still
 * unconverged counts after it.  A non-zero return leaves the loop (the solve ends like one that hit maxIter).  bench.py
 * times single iterations with it; fn == NULL removes the hook. */
typedef int (*chase_hip_iteration_fn)(void* user, size_t iteration, size_t filtered_vecs, size_t locked,
                                      size_t unconverged);
int chase_hip_solver_set_iteration_hook(chase_hip_solver* s, chase_hip_iteration_fn fn, void* user);
/* Scalar tape (chase_amd/host/tape.hpp).  The driver of chase_hip_solver_solve (algorithm/algorithm.inc:1376-1788) steers a
 * solve only by the host-visible numbers the Impl returns: Ritz values (RR), residuals (Resd), the Lanczos outputs.  mode 1:
 * the next solves record them (plus the QR variant each QR took and the number of residuals re-taken on the tolerance);
 * mode 2: the next solves replay a loaded tape - the Impl executes every operator at ITS shapes, the driver is shown the
 * recorded numbers and so issues exactly the recorded call sequence; 0: off.  With a loopback grid
 * (chase_hip_grid_create_loopback) this measures ONE rank of a multi-GPU solve on a one-GPU box (bench.py --replay-rank).
 * get keys after a replay: tape_qr_mismatches (QR calls that took another variant than recorded), tape_position, tape_size,
 * tape_tolerated (pseudo-Hermitian replay: projected matrices of partial sums that did not factorise, run on the identity).
 * Both drivers (chase::Solve and, on the grid Impl, chase::Solve_pseudo).  The data pointer stays valid until the next solve / load on this solver. */
int chase_hip_solver_tape_mode(chase_hip_solver* s, int mode);
int chase_hip_solver_tape_data(chase_hip_solver* s, const double** data, size_t* count);
int chase_hip_solver_tape_load(chase_hip_solver* s, const double* data, size_t count);

/* Algorithm<T>::lanczos_for_H2 (algorithm/algorithm.inc:1217-1373; tests/algorithm/lanczos_for_H2_test.cpp) on a
 * pseudo-Hermitian solver: DoS-based estimates of the H^2 spectrum go to the solver's ritzv[0 .. nev+nex), *upperb = b_sup,
 * *idx = number of Ritz directions moved into the start block */
int chase_hip_solver_lanczos_for_h2(chase_hip_solver* s, int numvec, int m, double* upperb, size_t* idx);
int chase_hip_solver_stats(chase_hip_solver* s, chase_hip_stats* out);
const double* chase_hip_solver_resid(chase_hip_solver* s); /* nev+nex residuals (host) */
const char* chase_hip_solver_trace(chase_hip_solver* s);   /* '\n'-separated virtual-call trace of the last solve */
/* resid[j] = || H v_j - lambda[j] v_j ||_2 for the first ncols vectors the Impl holds (after a solve: the eigenvectors),
 * from a fresh four-product H V - never from products an earlier step left behind.  The independent check the reference's
 * solve tests make after a solve (tests/chase_serial_solve.cpp:144-148,195-199, tests/chase_distributed_solve.cpp:209-284).
 * Collective over the grid for the distributed Impls. */
int chase_hip_solver_recompute_residuals(chase_hip_solver* s, size_t ncols, const double* lambda, double* resid);
int chase_hip_solver_peek_v(chase_hip_solver* s, chase_hip_ctx* ctx, void* host, size_t ldh);
/* chase_hip_hash64 of the first ncols columns of the Impl's (local) vector block, where it sits: the replicas of a block over
 * the grid columns are compared by 8 bytes per rank instead of by download */
int chase_hip_solver_hash_v(chase_hip_solver* s, chase_hip_ctx* ctx, size_t ncols, unsigned long long* hash);

/* the ChaseBase virtuals */
int chase_hip_op_start(chase_hip_solver* s);
int chase_hip_op_end(chase_hip_solver* s);
int chase_hip_op_initvecs(chase_hip_solver* s, int random);
/* optional hook ReinitColumns (algorithm/interface.hpp; chase_cpu.hpp:329-349, pchase_cpu.hpp:313-331) */
int chase_hip_op_reinit_columns(chase_hip_solver* s, size_t fixednev, const size_t* col_indices, size_t n_indices);
int chase_hip_op_shift(chase_hip_solver* s, double c, int isunshift);
int chase_hip_op_hemm(chase_hip_solver* s, size_t block, const double* alpha, const double* beta, size_t offset_left,
                      size_t offset_right);
int chase_hip_op_hemm_h2(chase_hip_solver* s, size_t block, const double* alpha, const double* beta, const double* gamma,
                         size_t offset_left, size_t offset_right);
int chase_hip_op_kconj(chase_hip_solver* s, size_t block);
int chase_hip_op_qr(chase_hip_solver* s, size_t fixednev, double cond);
int chase_hip_op_rr(chase_hip_solver* s, double* ritzv, size_t block);
int chase_hip_op_resd(chase_hip_solver* s, double* ritzv, double* resd, size_t fixednev);
int chase_hip_op_swap(chase_hip_solver* s, size_t i, size_t j);
int chase_hip_op_lock(chase_hip_solver* s, size_t new_converged);
/* numvec == 0 selects the single-vector Lanczos(m, upperb) overload */
int chase_hip_op_lanczos(chase_hip_solver* s, size_t M, size_t numvec, double* upperb, double* ritzv, double* Tau,
                         double* ritzV);
int chase_hip_op_lanczos_dos(chase_hip_solver* s, size_t idx, size_t m, void* ritzVc);
int chase_hip_op_check_symmetry(chase_hip_solver* s, int* is_sym);
/* symOrHermMatrix(uplo) (algorithm/interface.hpp:259): complete the Hermitian matrix from its stored triangle 'U' / 'L'.
 * Sequential Impl: on the caller's host copy (linalg/internal/cpu/symOrHerm.hpp); distributed Impl: in place on the device
 * shards, collective over the grid (linalg/internal/mpi/symOrHerm.hpp:127-320 without ScaLAPACK) */
int chase_hip_op_sym_or_herm(chase_hip_solver* s, char uplo);

#ifdef __cplusplus
}
#endif
#endif
