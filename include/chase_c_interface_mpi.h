/* chase_c_interface_mpi.h — the reference's distributed initialisation entry points with their EXACT signatures
 * (MPI_Comm* / MPI_Fint* communicator): interface/chase_c_interface.h:61-149 (block-cyclic and block layout, Hermitian and
 * pseudo-Hermitian, fp64 real / complex) and the *_f_ twins of interface/chase_c_interface.cpp:2425-3030.
 *
 * Exported by chase_amd/lib/libchase_hip_mpi.so (chase_amd/host/c_interface_mpi.c; built when an MPI installation is found).
 * Each builds the 2D grid from dim0 x dim1 on the communicator (grid_major 'C' / 'R'; grid/mpiGrid2D.hpp:189-446), one RCCL
 * communicator per grid row and per grid column (unique ids broadcast over MPI like grid/mpiGrid2D.hpp:448-484), the device
 * context of this rank, and then calls the grid-handle form of include/chase_c_interface.h (p?chase_init*_hip_).  Solve,
 * eigenpair read-back, finalize and the setters are the p?chase_* entry points of chase_c_interface.h (libchase_hip.so).
 * Single precision (ps / pc) is out of scope of this backend. */
#ifndef CHASE_C_INTERFACE_MPI_H
#define CHASE_C_INTERFACE_MPI_H
#include <mpi.h>
#include "chase_c_interface.h"

#ifdef __cplusplus
#include <complex>
typedef std::complex<double> chase_mpi_dcomplex;
extern "C" {
#else
typedef double _Complex chase_mpi_dcomplex;
#endif

void pdchase_init_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, double* V, double* ritzv,
    int* dim0, int* dim1, char* grid_major, MPI_Comm* comm, int* init);
void pdchase_init_internal_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, int* dim0, int* dim1,
    char* grid_major, MPI_Comm* comm, int* init);
void pzchase_init_(int* N, int* nev, int* nex, int* m, int* n, chase_mpi_dcomplex* H, int* ldh,
    chase_mpi_dcomplex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, MPI_Comm* comm, int* init);
void pzchase_init_internal_(int* N, int* nev, int* nex, int* m, int* n, chase_mpi_dcomplex* H, int* ldh, int* dim0,
    int* dim1, char* grid_major, MPI_Comm* comm, int* init);
void pzchase_init_pseudo_(int* N, int* nev, int* nex, int* m, int* n, chase_mpi_dcomplex* H, int* ldh,
    chase_mpi_dcomplex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, MPI_Comm* comm, int* init);
void pzchase_init_pseudo_internal_(int* N, int* nev, int* nex, int* m, int* n, chase_mpi_dcomplex* H, int* ldh,
    int* dim0, int* dim1, char* grid_major, MPI_Comm* comm, int* init);
void pdchase_init_blockcyclic_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh, double* V,
    double* ritzv, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Comm* comm, int* init);
void pdchase_init_blockcyclic_internal_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh,
    int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Comm* comm, int* init);
void pzchase_init_blockcyclic_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, chase_mpi_dcomplex* H, int* ldh,
    chase_mpi_dcomplex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc,
    MPI_Comm* comm, int* init);
void pzchase_init_blockcyclic_internal_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, chase_mpi_dcomplex* H,
    int* ldh, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Comm* comm, int* init);
void pzchase_init_pseudo_blockcyclic_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, chase_mpi_dcomplex* H,
    int* ldh, chase_mpi_dcomplex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc,
    MPI_Comm* comm, int* init);
void pzchase_init_pseudo_blockcyclic_internal_(int* N, int* nev, int* nex, int* mbsize, int* nbsize,
    chase_mpi_dcomplex* H, int* ldh, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Comm* comm,
    int* init);
void pdchase_init_f_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, double* V, double* ritzv,
    int* dim0, int* dim1, char* grid_major, MPI_Fint* fcomm, int* init);
void pdchase_init_internal_f_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, int* dim0, int* dim1,
    char* grid_major, MPI_Fint* fcomm, int* init);
void pzchase_init_f_(int* N, int* nev, int* nex, int* m, int* n, chase_mpi_dcomplex* H, int* ldh,
    chase_mpi_dcomplex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, MPI_Fint* fcomm, int* init);
void pzchase_init_internal_f_(int* N, int* nev, int* nex, int* m, int* n, chase_mpi_dcomplex* H, int* ldh, int* dim0,
    int* dim1, char* grid_major, MPI_Fint* fcomm, int* init);
void pzchase_init_pseudo_f_(int* N, int* nev, int* nex, int* m, int* n, chase_mpi_dcomplex* H, int* ldh,
    chase_mpi_dcomplex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, MPI_Fint* fcomm, int* init);
void pzchase_init_pseudo_internal_f_(int* N, int* nev, int* nex, int* m, int* n, chase_mpi_dcomplex* H, int* ldh,
    int* dim0, int* dim1, char* grid_major, MPI_Fint* fcomm, int* init);
void pdchase_init_blockcyclic_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh,
    double* V, double* ritzv, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Fint* fcomm,
    int* init);
void pdchase_init_blockcyclic_internal_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh,
    int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Fint* fcomm, int* init);
void pzchase_init_blockcyclic_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, chase_mpi_dcomplex* H,
    int* ldh, chase_mpi_dcomplex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc,
    MPI_Fint* fcomm, int* init);
void pzchase_init_blockcyclic_internal_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize,
    chase_mpi_dcomplex* H, int* ldh, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Fint* fcomm,
    int* init);
void pzchase_init_pseudo_blockcyclic_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, chase_mpi_dcomplex* H,
    int* ldh, chase_mpi_dcomplex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc,
    MPI_Fint* fcomm, int* init);
void pzchase_init_pseudo_blockcyclic_internal_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize,
    chase_mpi_dcomplex* H, int* ldh, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Fint* fcomm,
    int* init);

#ifdef __cplusplus
}
#endif
#endif
