/* c_dist_mpi.c — an MPI application in C on the DISTRIBUTED entry points of ChASE's C interface, served by
 * libchase_hip_mpi.so + libchase_hip.so (one MI355X per rank).  Same calls, in the same order, as an application written
 * against the reference's interface/chase_c_interface.h (p?chase_init_, p?chase_, p?chase_finalize_; the reference ships
 * such a driver as examples/4_interface/4_c_dist_chase.c): only the link line changes.
 *
 * The matrix is the Clement-type matrix of the reference's tests (eigenvalues -N, -N+2, ...), every rank fills its block of
 * the 2D block distribution on the host; the result is checked against the analytic spectrum and, on every rank, through the
 * residual of its local rows after an MPI reduction.
 *
 * build:  gcc -O2 -std=gnu11 -Iinclude -I$MPI_INC examples/c_dist_mpi.c -Lchase_amd/lib -lchase_hip_mpi -lchase_hip \
 *             $MPI_LIB/libmpi.so -Wl,--allow-shlib-undefined -Wl,--enable-new-dtags -Wl,-rpath,$PWD/chase_amd/lib \
 *             -Wl,-rpath,$MPI_LIB -lm
 *         (--allow-shlib-undefined: a conda MPI directory also holds an older libstdc++ that the linker must not check the
 *          ROCm libraries against; at run time every library resolves its own dependencies)
 * run:    mpiexec -n 4 ./a.out        (or ./a.out for one rank)
 */
#include <complex.h>
#include <math.h>
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>

/* the reference's prototypes (interface/chase_c_interface.h:126-128,177-195) */
void pzchase_init_(int* N, int* nev, int* nex, int* m, int* n, double _Complex* H, int* ldh, double _Complex* V, double* ritzv,
                   int* dim0, int* dim1, char* grid_major, MPI_Comm* comm, int* init);
void pzchase_(int* deg, double* tol, char* mode, char* opt, char* qr);
void pzchase_finalize_(int* flag);

static int block_len(int N, int p) { return N % p == 0 ? N / p : (N / p + 1 < N ? N / p + 1 : N); }

int main(int argc, char** argv)
{
    MPI_Init(&argc, &argv);
    int rank, size;
    MPI_Comm comm = MPI_COMM_WORLD;
    MPI_Comm_rank(comm, &rank);
    MPI_Comm_size(comm, &size);
    int N = argc > 1 ? atoi(argv[1]) : 1000, nev = 60, nex = 40, init = 0, flag = 1, deg = 20;
    double tol = 1e-10;
    char mode = 'R', opt = 'S', qr = 'C', major = 'C';
    int dims[2] = {0, 0};
    MPI_Dims_create(size, 2, dims);                          /* as square as possible, dims[0] >= dims[1] */
    const int myrow = rank % dims[0], mycol = rank / dims[0]; /* column-major grid */
    const int mb = block_len(N, dims[0]), nb = block_len(N, dims[1]);
    const int r0 = myrow * mb, c0 = mycol * nb;
    int m = r0 >= N ? 0 : (N - r0 < mb ? N - r0 : mb), n = c0 >= N ? 0 : (N - c0 < nb ? N - c0 : nb);
    double _Complex* H = calloc((size_t)m * n, sizeof *H);
    double _Complex* V = calloc((size_t)m * (nev + nex), sizeof *V);
    double* lambda = calloc((size_t)(nev + nex), sizeof *lambda);
    /* like the reference's example (examples/4_interface/4_c_dist_chase.c): the block is handed to init first and filled
     * afterwards - the interface keeps the pointer and reads the block at every pzchase_ call */
    pzchase_init_(&N, &nev, &nex, &m, &n, H, &m, V, lambda, &dims[0], &dims[1], &major, &comm, &init);
    if (!init) { fprintf(stderr, "rank %d: pzchase_init_ failed\n", rank); MPI_Abort(comm, 2); }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) {
            const int gi = r0 + i, gj = c0 + j;
            if (gi == gj + 1) H[i + (size_t)j * m] = sqrt((double)gj * (double)(N + 1 - gj));
            if (gj == gi + 1) H[i + (size_t)j * m] = sqrt((double)gi * (double)(N + 1 - gi));
        }
    pzchase_(&deg, &tol, &mode, &opt, &qr);
    /* analytic spectrum and the residual of the first pairs: r = H v - lambda v needs the whole vector -> gather it */
    double worst = 0.0;
    for (int k = 0; k < nev; ++k) worst = fmax(worst, fabs(lambda[k] - (-(double)N + 2.0 * k)));
    double _Complex* full = calloc((size_t)N, sizeof *full);
    double _Complex* acc = calloc((size_t)N, sizeof *acc);
    double resid_max = 0.0;
    for (int k = 0; k < 3; ++k) {
        for (int i = 0; i < N; ++i) full[i] = 0;
        if (mycol == 0) for (int i = 0; i < m; ++i) full[r0 + i] = V[i + (size_t)k * m];
        MPI_Allreduce(MPI_IN_PLACE, full, 2 * N, MPI_DOUBLE, MPI_SUM, comm);
        for (int i = 0; i < N; ++i) acc[i] = 0;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < m; ++i) acc[r0 + i] += H[i + (size_t)j * m] * full[c0 + j];
        MPI_Allreduce(MPI_IN_PLACE, acc, 2 * N, MPI_DOUBLE, MPI_SUM, comm);
        double s = 0.0;
        for (int i = 0; i < N; ++i) { const double _Complex d = acc[i] - lambda[k] * full[i]; s += creal(d) * creal(d) + cimag(d) * cimag(d); }
        resid_max = fmax(resid_max, sqrt(s));
    }
    pzchase_finalize_(&flag);
    if (rank == 0)
        printf("c_dist_mpi: %d rank(s), grid %d x %d, N = %d: lambda[0..2] = %.6f %.6f %.6f, max |lambda - exact| = %.2e, "
               "max residual = %.2e -> %s\n", size, dims[0], dims[1], N, lambda[0], lambda[1], lambda[2], worst, resid_max,
               (worst < 1e-8 && resid_max < 1e-8 && flag == 0) ? "OK" : "FAILED");
    const int ok = worst < 1e-8 && resid_max < 1e-8 && flag == 0;
    free(H); free(V); free(lambda); free(full); free(acc);
    MPI_Finalize();
    return ok ? 0 : 1;
}
