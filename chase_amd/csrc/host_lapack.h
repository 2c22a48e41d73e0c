// host_lapack.h — run-time bound host LAPACK provider (dlopen; LP64 Fortran ABI).
// The north star keeps the small dense eigenproblems on the host: HEEVD of the projected matrix
// (reference: lapackpp::t_heevd, linalg/internal/cpu/rayleighRitz.hpp:104; mpi/rayleighRitz.hpp:174) and the
// tridiagonal STEMR of Lanczos (reference: linalg/internal/cpu/lanczos.hpp:188).
#pragma once
#include <complex>
namespace chase_hip {
// returns 0 when a provider is bound; CHASE_HIP_ELAPACK otherwise (error text set)
int lapack_bind(const char* path_hint);
const char* lapack_provider();
// A (n x n, column-major, lda) Hermitian, lower triangle referenced; eigenvalues ascending in w, eigenvectors overwrite A
int host_heevd(bool cplx, int n, double* A, int lda, double* w);
// symmetric tridiagonal: all eigenpairs.  d[n], e[n] (e[n-1] workspace) are destroyed.  Z is n x n column-major.
int host_stemr(int n, double* d, double* e, double* w, double* Z, int ldz);
int host_stedc(int n, double* d, double* e, double* w, double* Z, int ldz);   // divide & conquer, same contract
// small dense core of the pseudo-Hermitian Rayleigh-Ritz (see host_lapack.cpp); A, M are n x n host, column-major
int host_pseudo_rr(bool cplx, int n, double* A, double* M, double* w);
int host_pseudo_rr_pre(bool cplx, int n, double* A, double* M);
int host_pseudo_rr_post(bool cplx, int n, const double* A, double* M, double* w);
void lapack_set_threads(int nthreads);
}
