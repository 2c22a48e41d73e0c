/* c_interface_mpi.c — the reference's distributed C entry points with their EXACT signatures (MPI_Comm* comm), on top of
 * the grid-handle forms of libchase_hip.so (chase_amd/host/c_interface_dist.cpp).  Built into chase_amd/lib/
 * libchase_hip_mpi.so only when an MPI installation is found (Makefile: MPI_INC / MPI_LIB); plain C so that it needs
 * nothing but libmpi and libchase_hip.
 *
 * Replaces interface/chase_c_interface.cpp:905-1290 (ChASE_DIST<...>::Initialize: MpiGrid2D from dim0 x dim1 and the
 * communicator, grid_major 'C' / 'R') and grid/mpiGrid2D.hpp:448-484 (one NCCL unique id per row / column communicator,
 * created by the group's first rank and broadcast over MPI).  One process per GPU: the device is CHASE_HIP_DEVICE if set,
 * else the rank inside the node (MPI_COMM_TYPE_SHARED) modulo the visible devices. */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/chase_c_interface_mpi.h"
#include "../../include/chase_hip.h"
#include "../../include/chase_hip_grid.h"

/* builds the context and the RCCL grid for this rank; returns NULL (and leaves *ctx NULL) on failure */
static chase_hip_grid* make_grid(int dim0, int dim1, const char* grid_major, MPI_Comm comm, chase_hip_ctx** ctx)
{
    int rank = 0, size = 0;
    *ctx = NULL;
    MPI_Comm_rank(comm, &rank);
    MPI_Comm_size(comm, &size);
    if (dim0 < 1 || dim1 < 1 || dim0 * dim1 != size || !grid_major || (*grid_major != 'C' && *grid_major != 'R')) {
        fprintf(stderr, "chase_hip: p?chase_init: dim0 x dim1 must equal the communicator size, grid_major 'C' or 'R'\n");
        return NULL;
    }
    /* coordinates: column-major rank = row + col*dim0, row-major rank = row*dim1 + col (grid/mpiGrid2D.hpp:402-432) */
    const int myrow = (*grid_major == 'C') ? rank % dim0 : rank / dim1;
    const int mycol = (*grid_major == 'C') ? rank / dim0 : rank % dim1;
    int dev = 0;
    const char* e = getenv("CHASE_HIP_DEVICE");
    if (e) dev = atoi(e);
    else {
        MPI_Comm node;
        int lrank = 0;
        MPI_Comm_split_type(comm, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);
        MPI_Comm_rank(node, &lrank);
        MPI_Comm_free(&node);
        /* the devices this process can see (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES respected); a launcher that already
         * binds one device per rank leaves exactly one, and every rank then takes device 0 */
        const char* vis = getenv("CHASE_HIP_DEVICES_PER_NODE");
        int ndev = vis ? atoi(vis) : chase_hip_device_count();
        dev = lrank % (ndev > 0 ? ndev : 1);
    }
    int ok = chase_hip_ctx_create(ctx, dev, NULL) == 0, all_ok = 0;
    /* nobody enters ncclCommInitRank unless every rank has its device (it blocks until all members arrive) */
    MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, comm);
    if (!all_ok) {
        if (*ctx) { chase_hip_ctx_destroy(*ctx); *ctx = NULL; }
        return NULL;
    }
    /* one unique id per row group and per column group, from the group's first member */
    MPI_Comm row_comm, col_comm;
    MPI_Comm_split(comm, myrow, mycol, &row_comm);       /* members of my grid row, ordered by column */
    MPI_Comm_split(comm, mycol, myrow, &col_comm);
    char id_row[CHASE_HIP_UNIQUE_ID_BYTES], id_col[CHASE_HIP_UNIQUE_ID_BYTES];
    memset(id_row, 0, sizeof id_row);
    memset(id_col, 0, sizeof id_col);
    if (mycol == 0) ok = ok && chase_hip_rccl_unique_id(id_row) == 0;
    if (myrow == 0) ok = ok && chase_hip_rccl_unique_id(id_col) == 0;
    MPI_Bcast(id_row, sizeof id_row, MPI_BYTE, 0, row_comm);
    MPI_Bcast(id_col, sizeof id_col, MPI_BYTE, 0, col_comm);
    MPI_Comm_free(&row_comm);
    MPI_Comm_free(&col_comm);
    MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, comm);
    chase_hip_grid* g = NULL;
    /* the grid object derives its coordinates column-major from the rank it is given */
    if (all_ok && chase_hip_grid_create_rccl(&g, *ctx, dim0, dim1, myrow + mycol * dim0, id_row, id_col) != 0) g = NULL;
    ok = g != NULL;
    MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, comm);
    if (!all_ok) {
        if (g) chase_hip_grid_destroy(g);
        chase_hip_ctx_destroy(*ctx);
        *ctx = NULL;
        fprintf(stderr, "chase_hip: p?chase_init: RCCL grid creation failed on rank %d: %s\n", rank, chase_hip_last_error());
        return NULL;
    }
    return g;
}

#define GRID_OR_FAIL                                                                                                   \
    chase_hip_ctx* ctx = NULL;                                                                                         \
    chase_hip_grid* g = make_grid(*dim0, *dim1, grid_major, *comm, &ctx);                                             \
    if (!g) { *init = 0; return; }                                                                                     \
    chase_hip_cshim_use_ctx(ctx, 1)

/* ---- block layout (interface/chase_c_interface.h:126-149) ---- */
void pdchase_init_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, double* V, double* ritzv, int* dim0,
                   int* dim1, char* grid_major, MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pdchase_init_hip_(N, nev, nex, m, n, H, ldh, V, ritzv, g, init);
}
void pdchase_init_internal_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, int* dim0, int* dim1,
                            char* grid_major, MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pdchase_init_internal_hip_(N, nev, nex, m, n, H, ldh, g, init);
}
void pzchase_init_(int* N, int* nev, int* nex, int* m, int* n, double _Complex* H, int* ldh, double _Complex* V,
                   double* ritzv, int* dim0, int* dim1, char* grid_major, MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pzchase_init_hip_(N, nev, nex, m, n, H, ldh, V, ritzv, g, init);
}
void pzchase_init_internal_(int* N, int* nev, int* nex, int* m, int* n, double _Complex* H, int* ldh, int* dim0, int* dim1,
                            char* grid_major, MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pzchase_init_internal_hip_(N, nev, nex, m, n, H, ldh, g, init);
}
void pzchase_init_pseudo_(int* N, int* nev, int* nex, int* m, int* n, double _Complex* H, int* ldh, double _Complex* V,
                          double* ritzv, int* dim0, int* dim1, char* grid_major, MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pzchase_init_pseudo_hip_(N, nev, nex, m, n, H, ldh, V, ritzv, g, init);
}
void pzchase_init_pseudo_internal_(int* N, int* nev, int* nex, int* m, int* n, double _Complex* H, int* ldh, int* dim0,
                                   int* dim1, char* grid_major, MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pzchase_init_pseudo_internal_hip_(N, nev, nex, m, n, H, ldh, g, init);
}
/* ---- block-cyclic layout (interface/chase_c_interface.h:61-124) ---- */
void pdchase_init_blockcyclic_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh, double* V,
                               double* ritzv, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc,
                               MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pdchase_init_blockcyclic_hip_(N, nev, nex, mbsize, nbsize, H, ldh, V, ritzv, irsrc, icsrc, g, init);
}
void pdchase_init_blockcyclic_internal_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh,
                                        int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Comm* comm,
                                        int* init)
{
    GRID_OR_FAIL;
    pdchase_init_blockcyclic_internal_hip_(N, nev, nex, mbsize, nbsize, H, ldh, irsrc, icsrc, g, init);
}
void pzchase_init_blockcyclic_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double _Complex* H, int* ldh,
                               double _Complex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, int* irsrc,
                               int* icsrc, MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pzchase_init_blockcyclic_hip_(N, nev, nex, mbsize, nbsize, H, ldh, V, ritzv, irsrc, icsrc, g, init);
}
void pzchase_init_blockcyclic_internal_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double _Complex* H, int* ldh,
                                        int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Comm* comm,
                                        int* init)
{
    GRID_OR_FAIL;
    pzchase_init_blockcyclic_internal_hip_(N, nev, nex, mbsize, nbsize, H, ldh, irsrc, icsrc, g, init);
}
void pzchase_init_pseudo_blockcyclic_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double _Complex* H, int* ldh,
                                      double _Complex* V, double* ritzv, int* dim0, int* dim1, char* grid_major,
                                      int* irsrc, int* icsrc, MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pzchase_init_pseudo_blockcyclic_hip_(N, nev, nex, mbsize, nbsize, H, ldh, V, ritzv, irsrc, icsrc, g, init);
}
void pzchase_init_pseudo_blockcyclic_internal_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double _Complex* H,
                                               int* ldh, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc,
                                               MPI_Comm* comm, int* init)
{
    GRID_OR_FAIL;
    pzchase_init_pseudo_blockcyclic_internal_hip_(N, nev, nex, mbsize, nbsize, H, ldh, irsrc, icsrc, g, init);
}

/* ---- Fortran communicators: the reference's *_f_ twins (interface/chase_c_interface.cpp:2425-3030) take an MPI_Fint handle,
 * convert it with MPI_Comm_f2c and do the same ---- */
void pdchase_init_f_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, double* V, double* ritzv, int*
                     dim0, int* dim1, char* grid_major, MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pdchase_init_(N, nev, nex, m, n, H, ldh, V, ritzv, dim0, dim1, grid_major, &comm, init);
}
void pdchase_init_internal_f_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, int* dim0, int* dim1,
                              char* grid_major, MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pdchase_init_internal_(N, nev, nex, m, n, H, ldh, dim0, dim1, grid_major, &comm, init);
}
void pzchase_init_f_(int* N, int* nev, int* nex, int* m, int* n, double _Complex* H, int* ldh, double _Complex* V,
                     double* ritzv, int* dim0, int* dim1, char* grid_major, MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pzchase_init_(N, nev, nex, m, n, H, ldh, V, ritzv, dim0, dim1, grid_major, &comm, init);
}
void pzchase_init_internal_f_(int* N, int* nev, int* nex, int* m, int* n, double _Complex* H, int* ldh, int* dim0,
                              int* dim1, char* grid_major, MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pzchase_init_internal_(N, nev, nex, m, n, H, ldh, dim0, dim1, grid_major, &comm, init);
}
void pzchase_init_pseudo_f_(int* N, int* nev, int* nex, int* m, int* n, double _Complex* H, int* ldh, double _Complex*
                            V, double* ritzv, int* dim0, int* dim1, char* grid_major, MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pzchase_init_pseudo_(N, nev, nex, m, n, H, ldh, V, ritzv, dim0, dim1, grid_major, &comm, init);
}
void pzchase_init_pseudo_internal_f_(int* N, int* nev, int* nex, int* m, int* n, double _Complex* H, int* ldh, int*
                                     dim0, int* dim1, char* grid_major, MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pzchase_init_pseudo_internal_(N, nev, nex, m, n, H, ldh, dim0, dim1, grid_major, &comm, init);
}
void pdchase_init_blockcyclic_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh, double* V,
                                 double* ritzv, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc,
                                 MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pdchase_init_blockcyclic_(N, nev, nex, mbsize, nbsize, H, ldh, V, ritzv, dim0, dim1, grid_major, irsrc, icsrc,
        &comm, init);
}
void pdchase_init_blockcyclic_internal_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh,
                                          int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc, MPI_Fint*
                                          fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pdchase_init_blockcyclic_internal_(N, nev, nex, mbsize, nbsize, H, ldh, dim0, dim1, grid_major, irsrc, icsrc,
        &comm, init);
}
void pzchase_init_blockcyclic_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double _Complex* H, int* ldh,
                                 double _Complex* V, double* ritzv, int* dim0, int* dim1, char* grid_major, int*
                                 irsrc, int* icsrc, MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pzchase_init_blockcyclic_(N, nev, nex, mbsize, nbsize, H, ldh, V, ritzv, dim0, dim1, grid_major, irsrc, icsrc,
        &comm, init);
}
void pzchase_init_blockcyclic_internal_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double _Complex* H,
                                          int* ldh, int* dim0, int* dim1, char* grid_major, int* irsrc, int* icsrc,
                                          MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pzchase_init_blockcyclic_internal_(N, nev, nex, mbsize, nbsize, H, ldh, dim0, dim1, grid_major, irsrc, icsrc,
        &comm, init);
}
void pzchase_init_pseudo_blockcyclic_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double _Complex* H, int*
                                        ldh, double _Complex* V, double* ritzv, int* dim0, int* dim1, char*
                                        grid_major, int* irsrc, int* icsrc, MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pzchase_init_pseudo_blockcyclic_(N, nev, nex, mbsize, nbsize, H, ldh, V, ritzv, dim0, dim1, grid_major, irsrc,
        icsrc, &comm, init);
}
void pzchase_init_pseudo_blockcyclic_internal_f_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double
                                                 _Complex* H, int* ldh, int* dim0, int* dim1, char* grid_major, int*
                                                 irsrc, int* icsrc, MPI_Fint* fcomm, int* init)
{
    MPI_Comm comm = MPI_Comm_f2c(*fcomm);
    pzchase_init_pseudo_blockcyclic_internal_(N, nev, nex, mbsize, nbsize, H, ldh, dim0, dim1, grid_major, irsrc,
        icsrc, &comm, init);
}
