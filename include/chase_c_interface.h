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
 * The GPU is chosen by CHASE_HIP_DEVICE (default 0).  Single precision and the distributed p?chase_* entry points (they
 * take an MPI_Comm; this image has no MPI) are not provided by this fp64, single-process shim — the distributed Impl is
 * reached through chase_hip_solver.h. */
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
#ifdef __cplusplus
}
#endif
#endif
