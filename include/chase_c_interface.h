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
 * The GPU is chosen by CHASE_HIP_DEVICE (default 0).  single precision and the distributed p?chase_* entry points are
 * not provided by this fp64, single-process shim (the distributed Impl is reached through chase_hip_solver.h). */
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
#ifdef __cplusplus
}
#endif
#endif
