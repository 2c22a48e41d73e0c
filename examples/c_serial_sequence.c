/* c_serial_sequence.c — a plain C caller of ChASE's application-facing interface (include/chase_c_interface.h, the same
 * entry points as the reference's interface/chase_c_interface.h), served by libchase_hip.so on an MI355X.
 *
 * Mirrors the life cycle of the reference's examples/4_interface/4_c_serial_chase.c: zchase_init_ is called once with the
 * (still empty) matrix buffer, the caller then fills / perturbs H in place and calls zchase_ repeatedly — the first solve
 * from random vectors (mode 'R'), the following ones from the previous eigenvectors (mode 'A', "sequence of eigenproblems").
 * Every solve is checked here by recomputing ||H v - lambda v|| for the first eigenpairs on the host.
 *
 * build:  gcc -O2 -std=c11 -Iinclude examples/c_serial_sequence.c -Lchase_amd/lib -lchase_hip -Wl,-rpath,$PWD/chase_amd/lib -lm
 */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "chase_c_interface.h"

static unsigned long long lcg_state = 88172645463325252ull;
static double lcg_uniform(void)            /* (0, 1) */
{
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return ((double)(lcg_state >> 11) + 0.5) / 9007199254740992.0;
}

int main(int argc, char** argv)
{
    int N = argc > 1 ? atoi(argv[1]) : 600, nev = 40, nex = 24, problems = 3;
    int deg = 20, init = 0, flag = 1;
    double tol = 1e-10, perturb = 1e-4;
    char mode = 'R', opt = 'S', qr = 'C';
    double _Complex* H = calloc((size_t)N * N, sizeof *H);
    double _Complex* V = calloc((size_t)N * (nev + nex), sizeof *V);
    double* lambda = calloc((size_t)(nev + nex), sizeof *lambda);
    if (!H || !V || !lambda) return 2;

    zchase_init_(&N, &nev, &nex, H, &N, V, lambda, &init);
    if (!init) { fprintf(stderr, "zchase_init_ failed\n"); return 3; }

    for (int i = 0; i + 1 < N; ++i) {                       /* Clement-type matrix of the reference's examples */
        const double v = sqrt((double)i * (double)(N + 1 - i));
        H[i + 1 + (size_t)N * i] = v;
        H[i + (size_t)N * (i + 1)] = v;
    }

    int bad = 0;
    for (int p = 0; p < problems; ++p) {
        zchase_(&deg, &tol, &mode, &opt, &qr);
        double worst = 0.0;
        for (int j = 0; j < 5; ++j) {                        /* recompute a few residuals like the reference's tests do */
            double r2 = 0.0;
            for (int i = 0; i < N; ++i) {
                double _Complex s = -lambda[j] * V[i + (size_t)N * j];
                for (int k = 0; k < N; ++k) s += H[i + (size_t)N * k] * V[k + (size_t)N * j];
                r2 += creal(s) * creal(s) + cimag(s) * cimag(s);
            }
            if (sqrt(r2) > worst) worst = sqrt(r2);
        }
        printf("problem %d (mode %c): lambda[0..2] = %.10f %.10f %.10f, worst recomputed residual %.3e\n", p, mode,
               lambda[0], lambda[1], lambda[2], worst);
        if (!(worst < 1e-8)) bad = 1;
        for (int i = 1; i < N; ++i)                           /* Hermitian perturbation, next problem of the sequence */
            for (int j = 1; j < i; ++j) {
                const double _Complex e = perturb * ((lcg_uniform() - 0.5) + (lcg_uniform() - 0.5) * I);
                H[j + (size_t)N * i] += e;
                H[i + (size_t)N * j] += conj(e);
            }
        mode = 'A';
    }
    zchase_finalize_(&flag);
    free(H); free(V); free(lambda);
    if (bad || flag) { printf("FAILED\n"); return 1; }   /* finalize reports 0 like the reference */
    printf("C_SEQUENCE_OK\n");
    return 0;
}
