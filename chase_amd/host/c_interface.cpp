// c_interface.cpp — ChASE's application-facing C interface (interface/chase_c_interface.h:13-41) on top of the HIP Impl.
#include <cstdlib>
#include "../../include/chase_c_interface.h"
#include "../../include/chase_hip.h"
#include "../../include/chase_hip_solver.h"

namespace {
chase_hip_ctx* g_ctx = nullptr;
chase_hip_solver* g_d = nullptr;
chase_hip_solver* g_z = nullptr;

bool ensure_ctx()
{
    if (g_ctx) return true;
    int dev = 0;
    if (const char* e = std::getenv("CHASE_HIP_DEVICE")) dev = std::atoi(e);
    return chase_hip_ctx_create(&g_ctx, dev, nullptr) == 0;
}
void init_common(chase_hip_solver** slot, int cplx, int N, int nev, int nex, void* H, int ldh, void* V, double* ritzv,
                 int* init)
{
    *init = 0;
    if (!ensure_ctx()) return;
    if (*slot) { chase_hip_solver_destroy(*slot); *slot = nullptr; }
    if (chase_hip_solver_create(slot, g_ctx, cplx, (size_t)N, (size_t)nev, (size_t)nex, H, (size_t)ldh, V, (size_t)N,
                                ritzv, 0) == 0)
        *init = 1;
}
void solve_common(chase_hip_solver* s, int deg, double tol, char mode, char opt, char qr)
{
    if (!s) return;
    chase_hip_solver_set(s, "tol", tol);
    chase_hip_solver_set(s, "deg", (double)deg);
    chase_hip_solver_set(s, "opt", opt == 'S' ? 1.0 : 0.0);
    chase_hip_solver_set(s, "approx", mode == 'A' ? 1.0 : 0.0);
    chase_hip_solver_set(s, "cholqr", qr == 'C' ? 1.0 : 0.0);
    chase_hip_solver_solve(s, 0);
}
} // namespace

extern "C" {
void dchase_init_(int* N, int* nev, int* nex, double* H, int* ldh, double* V, double* ritzv, int* init)
{
    init_common(&g_d, 0, *N, *nev, *nex, H, *ldh, V, ritzv, init);
}
void dchase_(int* deg, double* tol, char* mode, char* opt, char* qr) { solve_common(g_d, *deg, *tol, *mode, *opt, *qr); }
void dchase_finalize_(int* flag)
{
    if (g_d) { chase_hip_solver_destroy(g_d); g_d = nullptr; }
    *flag = 1;
}
void zchase_init_(int* N, int* nev, int* nex, void* H, int* ldh, void* V, double* ritzv, int* init)
{
    init_common(&g_z, 1, *N, *nev, *nex, H, *ldh, V, ritzv, init);
}
void zchase_(int* deg, double* tol, char* mode, char* opt, char* qr) { solve_common(g_z, *deg, *tol, *mode, *opt, *qr); }
void zchase_finalize_(int* flag)
{
    if (g_z) { chase_hip_solver_destroy(g_z); g_z = nullptr; }
    *flag = 1;
}
}
