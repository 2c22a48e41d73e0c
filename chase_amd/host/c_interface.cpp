// c_interface.cpp — ChASE's application-facing C interface (interface/chase_c_interface.h:13-58,177-181; sequential fp64
// entry points) on top of the HIP Impl.  Behaviour follows interface/chase_c_interface.cpp: one solver object per type
// kept between init / solve / finalize; zchase_ dispatches to the pseudo-Hermitian solver when that is the one that was
// initialised (chase_c_interface.cpp:2204-2220); the *_internal_ inits own the vector / Ritz-value storage and
// ?chase_get_eigenpairs_ copies the first nev eigenvectors and Ritz values out (chase_c_interface.cpp:2329-2400).
#include <dlfcn.h>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/chase_c_interface.h"
#include "../../include/chase_hip.h"
#include "../../include/chase_hip_solver.h"

namespace {
chase_hip_ctx* g_ctx = nullptr;

struct Slot {
    chase_hip_solver* s = nullptr;
    int cplx = 0, pseudo = 0;
    std::size_t N = 0, nev = 0, nex = 0;
    void* V = nullptr;                       // vector block the solver works on (caller's or internal)
    double* ritzv = nullptr;
    std::vector<double> own_v, own_ritzv;    // storage of the *_internal_ variants
    void clear()
    {
        if (s) { chase_hip_solver_destroy(s); s = nullptr; }
        own_v.clear(); own_v.shrink_to_fit(); own_ritzv.clear();
        V = nullptr; ritzv = nullptr;
    }
};
Slot g_d, g_z, g_zp;
int g_sym_check = 1;

bool ensure_ctx()
{
    if (g_ctx) return true;
    int dev = 0;
    if (const char* e = std::getenv("CHASE_HIP_DEVICE")) dev = std::atoi(e);
    return chase_hip_ctx_create(&g_ctx, dev, nullptr) == 0;
}
void init_common(Slot& sl, int cplx, int pseudo, int N, int nev, int nex, void* H, int ldh, void* V, double* ritzv, int* init)
{
    *init = 0;
    if (!ensure_ctx()) return;
    if (sl.s) { chase_hip_solver_destroy(sl.s); sl.s = nullptr; }
    const std::size_t ncol = (pseudo ? 2 : 1) * (std::size_t)(nev + nex);
    if (!V) {                                 // *_internal_: the interface owns V and ritzv
        sl.own_v.assign((std::size_t)N * ncol * (cplx ? 2 : 1), 0.0);
        sl.own_ritzv.assign(ncol, 0.0);
        V = sl.own_v.data(); ritzv = sl.own_ritzv.data();
    }
    sl.cplx = cplx; sl.pseudo = pseudo; sl.N = (std::size_t)N; sl.nev = (std::size_t)nev; sl.nex = (std::size_t)nex;
    sl.V = V; sl.ritzv = ritzv;
    const int rc = pseudo ? chase_hip_solver_create_pseudo(&sl.s, g_ctx, cplx, (size_t)N, (size_t)nev, (size_t)nex, H,
                                                           (size_t)ldh, V, (size_t)N, ritzv, 0)
                          : chase_hip_solver_create(&sl.s, g_ctx, cplx, (size_t)N, (size_t)nev, (size_t)nex, H, (size_t)ldh,
                                                    V, (size_t)N, ritzv, 0);
    if (rc == 0) *init = 1;
}
void solve_common(Slot& sl, int deg, double tol, char mode, char opt, char qr)
{
    if (!sl.s) return;
    chase_hip_solver_set(sl.s, "tol", tol);
    chase_hip_solver_set(sl.s, "deg", (double)deg);
    chase_hip_solver_set(sl.s, "opt", opt == 'S' ? 1.0 : 0.0);
    chase_hip_solver_set(sl.s, "approx", mode == 'A' ? 1.0 : 0.0);
    chase_hip_solver_set(sl.s, "cholqr", qr == 'C' ? 1.0 : 0.0);
    chase_hip_solver_solve(sl.s, 0);
}
// first nev eigenvectors (N x nev, leading dimension ld) and Ritz values
void get_pairs(const Slot& sl, void* out, int ld, double* ritzv)
{
    if (!sl.s || !sl.V || !out || ld < (int)sl.N) return;
    const std::size_t es = sl.cplx ? 16 : 8;
    for (std::size_t j = 0; j < sl.nev; ++j)
        std::memcpy((char*)out + j * (std::size_t)ld * es, (const char*)sl.V + j * sl.N * es, sl.N * es);
    if (ritzv) std::memcpy(ritzv, sl.ritzv, sl.nev * sizeof(double));
}
} // namespace

extern "C" {
void dchase_init_(int* N, int* nev, int* nex, double* H, int* ldh, double* V, double* ritzv, int* init)
{
    init_common(g_d, 0, 0, *N, *nev, *nex, H, *ldh, V, ritzv, init);
}
void dchase_init_internal_(int* N, int* nev, int* nex, double* H, int* ldh, int* init)
{
    init_common(g_d, 0, 0, *N, *nev, *nex, H, *ldh, nullptr, nullptr, init);
}
void dchase_(int* deg, double* tol, char* mode, char* opt, char* qr) { solve_common(g_d, *deg, *tol, *mode, *opt, *qr); }
void dchase_get_eigenpairs_(double* LEigsV, int* ld, double* ritzv) { if (ld) get_pairs(g_d, LEigsV, *ld, ritzv); }
void dchase_finalize_(int* flag)
{
    g_d.clear();
    if (flag) *flag = 0;                     // ChASE_SEQ<...>::Finalize() returns 0 (chase_c_interface.cpp:320-368)
}

void zchase_init_(int* N, int* nev, int* nex, void* H, int* ldh, void* V, double* ritzv, int* init)
{
    g_zp.clear();
    init_common(g_z, 1, 0, *N, *nev, *nex, H, *ldh, V, ritzv, init);
}
void zchase_init_internal_(int* N, int* nev, int* nex, void* H, int* ldh, int* init)
{
    g_zp.clear();
    init_common(g_z, 1, 0, *N, *nev, *nex, H, *ldh, nullptr, nullptr, init);
}
void zchase_init_pseudo_(int* N, int* nev, int* nex, void* H, int* ldh, void* V, double* ritzv, int* init)
{
    g_z.clear();
    init_common(g_zp, 1, 1, *N, *nev, *nex, H, *ldh, V, ritzv, init);
}
void zchase_init_pseudo_internal_(int* N, int* nev, int* nex, void* H, int* ldh, int* init)
{
    g_z.clear();
    init_common(g_zp, 1, 1, *N, *nev, *nex, H, *ldh, nullptr, nullptr, init);
}
void zchase_pseudo_(int* deg, double* tol, char* mode, char* opt, char* qr) { solve_common(g_zp, *deg, *tol, *mode, *opt, *qr); }
void zchase_(int* deg, double* tol, char* mode, char* opt, char* qr)
{
    if (g_zp.s) solve_common(g_zp, *deg, *tol, *mode, *opt, *qr);       // the type that was initialised decides
    else solve_common(g_z, *deg, *tol, *mode, *opt, *qr);
}
void zchase_get_eigenpairs_(void* LEigsV, int* ld, double* ritzv)
{
    if (!ld) return;
    if (g_zp.s) get_pairs(g_zp, LEigsV, *ld, ritzv);
    else get_pairs(g_z, LEigsV, *ld, ritzv);
}
void zchase_finalize_(int* flag)
{
    g_z.clear(); g_zp.clear();
    if (flag) *flag = 0;
}
void chase_enable_sym_check_(int* flag) { if (flag) g_sym_check = *flag != 0; }

/* ---- unified configuration setters and build queries (interface/chase_c_interface.h:207-238,
 * interface/chase_c_interface.cpp:3795-4260): they act on whichever solver instance is live - sequential or distributed -
 * and return silently when none is.  tol / deg / opt / approx / cholqr are also arguments of ?chase_ / p?chase_, which set
 * them again at every solve (chase_c_interface.cpp:444-466,1873-1886); the others persist across solves. */
chase_hip_solver* chase_hip_cshim_dist_solver(int cplx);
chase_hip_solver* chase_hip_cshim_seq_solver(int kind) { return kind == 0 ? g_d.s : kind == 1 ? g_z.s : kind == 2 ? g_zp.s : nullptr; }
static void set_all(const char* key, double v)
{
    chase_hip_solver* live[5] = {g_d.s, g_z.s, g_zp.s, chase_hip_cshim_dist_solver(0), chase_hip_cshim_dist_solver(1)};
    for (chase_hip_solver* s : live)
        if (s) chase_hip_solver_set(s, key, v);
}
void chase_set_tol_(double* v) { if (v) set_all("tol", *v); }
void chase_set_deg_(int* v) { if (v) set_all("deg", *v); }
void chase_set_max_deg_(int* v) { if (v) set_all("maxdeg", *v); }
void chase_set_deg_extra_(int* v) { if (v) set_all("degextra", *v); }
void chase_set_max_iter_(int* v) { if (v) set_all("maxiter", *v); }
void chase_set_lanczos_iter_(int* v) { if (v) set_all("lanczositer", *v); }
void chase_set_num_lanczos_(int* v) { if (v) set_all("numlanczos", *v); }
void chase_set_approx_(int* v) { if (v) set_all("approx", *v != 0); }
void chase_set_opt_(int* v) { if (v) set_all("opt", *v != 0); }
void chase_set_cholqr_(int* v) { if (v) set_all("cholqr", *v != 0); }
void chase_set_decaying_rate_(float* v) { if (v) set_all("decayingrate", *v); }
void chase_set_cluster_aware_degrees_(int* v) { if (v) set_all("clusteraware", *v != 0); }
void chase_set_upperb_scale_rate_(float* v) { if (v) set_all("upperbscale", *v); }

void chase_get_version_(char* version, int* len)
{
    if (!version || !len || *len <= 0) return;
    const char* ver = chase_hip_version();
    int n = 0;
    while (ver[n] != '\0' && n < *len - 1) { version[n] = ver[n]; ++n; }
    version[n] = '\0';
    *len = n;
}
void chase_has_cuda_(int* flag) { if (flag) *flag = 0; }          /* HIP on gfx950, no CUDA anywhere */
void chase_has_nccl_(int* flag) { if (flag) *flag = 1; }          /* the NCCL API, served by RCCL over xGMI */
void chase_has_scalapack_(int* flag) { if (flag) *flag = 0; }     /* own distributed Householder QR instead */
void chase_has_mpi_(int* flag)                                     /* 1 when the MPI front end (libchase_hip_mpi.so) is loaded */
{
    if (flag) *flag = dlsym(RTLD_DEFAULT, "pzchase_init_blockcyclic_") != nullptr ? 1 : 0;
}
void chase_print_config_()
{
    int mpi = 0;
    chase_has_mpi_(&mpi);
    std::printf("========================================\nChASE MI355X backend configuration\n========================================\n"
                "Version: %s\n  HIP / gfx950 kernels:    ENABLED\n  CUDA:                    DISABLED\n"
                "  RCCL (NCCL API):         ENABLED\n  MPI front end:           %s\n  ScaLAPACK:               DISABLED\n"
                "  host LAPACK provider:    %s\n", chase_hip_version(), mpi ? "LOADED" : "not loaded", chase_hip_lapack_provider());
    std::fflush(stdout);
}
/* aliases without the leading 'p' (interface/chase_c_interface.h:197-205): forward to p?chase_readHam_ */
void pdchase_readHam_(const char* filename);
void pzchase_readHam_(const char* filename);
void dchase_readHam_(const char* filename) { pdchase_readHam_(filename); }
void zchase_readHam_(const char* filename) { pzchase_readHam_(filename); }
}
