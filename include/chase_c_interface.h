/* chase_c_interface.h — the application-facing C / Fortran-style entry points of ChASE for the sequential fp64 paths,
 * served by the MI355X backend.  Same names, argument order and meaning as the reference's
 * interface/chase_c_interface.h:13-41 (dchase_init_, dchase_, dchase_finalize_ and the z variants; all arguments by
 * pointer, Fortran convention), so an application linked against ChASE's C interface (FLEUR / YAMBO style callers,
 * examples/4_interface) relinks against libchase_hip.so unchanged.
 *   init     : N, nev, nex, H (N x N, ldh, host), V (N x (nev+nex), host, may hold approximate vectors), ritzv;
 *              *init = 1 on success, 0 on failure (chase_hip_last_error() has the text)
 *   solve    : deg, tol, mode ('R' random start / 'A' approximate vectors in V), opt ('S' optimise degrees / 'X' not),
 *              qr ('C' CholQR / 'H' Householder)            (interface/chase_c_interface.cpp:444-466)
 *   finalize : *flag = 1
 *   *_internal_      : the interface owns V and ritzv (interface/chase_c_interface.h:25-32); read the result with
 *   ?chase_get_eigenpairs_(LEigsV, ld, ritzv): first nev eigenvectors (N x nev, ld >= N) and Ritz values (:177-181)
 *   zchase_init_pseudo_[internal_] / zchase_pseudo_ : pseudo-Hermitian (Bethe-Salpeter) problems through chase::Solve_pseudo;
 *              V has 2*(nev+nex) columns, ritzv 2*(nev+nex) entries (:44-58); zchase_ solves whichever type was initialised
 * The GPU is chosen by CHASE_HIP_DEVICE (default 0).  Single precision is not provided (this backend is fp64).
 *
 * Distributed entry points (interface/chase_c_interface.h:61-65,95-99,126-128,149,177-195): every p?chase_init* of the
 * reference takes an MPI_Comm* and builds the 2D grid from it.  libchase_hip.so exports the same entry points with the
 * communicator replaced by a chase_hip_grid* (suffix _hip_; the grid carries dims, coordinates and the RCCL row / column
 * communicators; tell the shim the grid's device context first: chase_hip_cshim_use_ctx), and libchase_hip_mpi.so (built
 * when mpi.h is found; chase_amd/host/c_interface_mpi.c) exports the reference's exact MPI_Comm* signatures on top of
 * them.  H is the caller's HOST block of the local rows / columns (m x n, ldh); V the host block of local rows
 * (m x (nev+nex), ld = m; 2*(nev+nex) columns for pseudo-Hermitian problems), read when mode == 'A' and written by the
 * solve; mbsize / nbsize select the block-cyclic layout, irsrc / icsrc must be 0.  After init the names are the
 * reference's: p?chase_, p?chase_get_eigenpairs_ (local rows of the first nev vectors), p?chase_finalize_,
 * p?chase_readHam_ / p?chase_wrtHam_ (the local shard from / to a raw column-major N x N file). */
#ifndef CHASE_C_INTERFACE_HIP_H
#define CHASE_C_INTERFACE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif
void dchase_init_(int* N, int* nev, int* nex, double* H, int* ldh, double* V, double* ritzv, int* init);
void dchase_(int* deg, double* tol, char* mode, char* opt, char* qr);
void dchase_finalize_(int* flag);
void zchase_init_(int* N, int* nev, int* nex, void* H, int* ldh, void* V, double* ritzv, int* init);
void zchase_(int* deg, double* tol, char* mode, char* opt, char* qr);
void zchase_finalize_(int* flag);
void dchase_init_internal_(int* N, int* nev, int* nex, double* H, int* ldh, int* init);
void zchase_init_internal_(int* N, int* nev, int* nex, void* H, int* ldh, int* init);
void zchase_init_pseudo_(int* N, int* nev, int* nex, void* H, int* ldh, void* V, double* ritzv, int* init);
void zchase_init_pseudo_internal_(int* N, int* nev, int* nex, void* H, int* ldh, int* init);
void zchase_pseudo_(int* deg, double* tol, char* mode, char* opt, char* qr);
void dchase_get_eigenpairs_(double* LEigsV, int* ld, double* ritzv);
void zchase_get_eigenpairs_(void* LEigsV, int* ld, double* ritzv);
void chase_enable_sym_check_(int* flag);     /* interface/chase_c_interface.cpp:4055: flag kept for callers that set it */
/* unified configuration setters and build queries (interface/chase_c_interface.h:207-238): act on the live solver instance
 * (sequential or distributed), silent when none is initialised */
void chase_set_tol_(double* tol);
void chase_set_deg_(int* deg);
void chase_set_max_deg_(int* max_deg);
void chase_set_deg_extra_(int* deg_extra);
void chase_set_max_iter_(int* max_iter);
void chase_set_lanczos_iter_(int* lanczos_iter);
void chase_set_num_lanczos_(int* num_lanczos);
void chase_set_approx_(int* flag);
void chase_set_opt_(int* flag);
void chase_set_cholqr_(int* flag);
void chase_set_decaying_rate_(float* decaying_rate);
void chase_set_cluster_aware_degrees_(int* flag);
void chase_set_upperb_scale_rate_(float* upperb_scale_rate);
void chase_get_version_(char* version, int* len);
void chase_has_cuda_(int* flag);          /* 0 */
void chase_has_nccl_(int* flag);          /* 1: the NCCL API is served by RCCL */
void chase_has_scalapack_(int* flag);     /* 0 */
void chase_has_mpi_(int* flag);           /* 1 when libchase_hip_mpi.so is loaded in the process */
void chase_print_config_(void);
void dchase_readHam_(const char* filename);  /* aliases of p?chase_readHam_ */
void zchase_readHam_(const char* filename);

/* ---- distributed (one process per GPU) ---- */
struct chase_hip_grid;
struct chase_hip_ctx;
struct chase_hip_solver;
int chase_hip_cshim_use_ctx(struct chase_hip_ctx* ctx, int own); /* context of the NEXT _hip_ init; own: finalize destroys grid + ctx */
struct chase_hip_solver* chase_hip_cshim_dist_solver(int cplx); /* the live distributed solver (chase_hip_solver.h), or NULL */
/* 1: every thread that calls a p?chase_init*_hip_ entry point from now on is one RANK of a grid living in this process (one
 * thread per GPU) with a solver of its own; 0 (default): ONE solver per type in the process, replaced by the next init from
 * any thread - the reference's static members (interface/chase_c_interface.cpp:905-1290).  Returns the previous setting. */
int chase_hip_cshim_thread_ranks(int on);
struct chase_hip_solver* chase_hip_cshim_seq_solver(int kind); /* live sequential solver: 0 real, 1 complex, 2 complex pseudo-Hermitian */
void pdchase_init_hip_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, double* V, double* ritzv,
                       struct chase_hip_grid* grid, int* init);
void pdchase_init_internal_hip_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh,
                                struct chase_hip_grid* grid, int* init);
void pzchase_init_hip_(int* N, int* nev, int* nex, int* m, int* n, void* H, int* ldh, void* V, double* ritzv,
                       struct chase_hip_grid* grid, int* init);
void pzchase_init_internal_hip_(int* N, int* nev, int* nex, int* m, int* n, void* H, int* ldh,
                                struct chase_hip_grid* grid, int* init);
void pzchase_init_pseudo_hip_(int* N, int* nev, int* nex, int* m, int* n, void* H, int* ldh, void* V, double* ritzv,
                              struct chase_hip_grid* grid, int* init);
void pzchase_init_pseudo_internal_hip_(int* N, int* nev, int* nex, int* m, int* n, void* H, int* ldh,
                                       struct chase_hip_grid* grid, int* init);
void pdchase_init_blockcyclic_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh, double* V,
                                   double* ritzv, int* irsrc, int* icsrc, struct chase_hip_grid* grid, int* init);
void pdchase_init_blockcyclic_internal_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh,
                                            int* irsrc, int* icsrc, struct chase_hip_grid* grid, int* init);
void pzchase_init_blockcyclic_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, void* H, int* ldh, void* V,
                                   double* ritzv, int* irsrc, int* icsrc, struct chase_hip_grid* grid, int* init);
void pzchase_init_blockcyclic_internal_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, void* H, int* ldh,
                                            int* irsrc, int* icsrc, struct chase_hip_grid* grid, int* init);
void pzchase_init_pseudo_blockcyclic_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, void* H, int* ldh, void* V,
                                          double* ritzv, int* irsrc, int* icsrc, struct chase_hip_grid* grid, int* init);
void pzchase_init_pseudo_blockcyclic_internal_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, void* H, int* ldh,
                                                   int* irsrc, int* icsrc, struct chase_hip_grid* grid, int* init);
void pdchase_(int* deg, double* tol, char* mode, char* opt, char* qr);
void pzchase_(int* deg, double* tol, char* mode, char* opt, char* qr);
void pdchase_get_eigenpairs_(double* LEigsV, int* ld, double* ritzv);
void pzchase_get_eigenpairs_(void* LEigsV, int* ld, double* ritzv);
void pdchase_finalize_(int* flag);
void pzchase_finalize_(int* flag);
void pdchase_readHam_(const char* filename);
void pzchase_readHam_(const char* filename);
void pdchase_wrtHam_(const char* filename);
void pzchase_wrtHam_(const char* filename);
#ifdef __cplusplus
}
#endif
#endif
