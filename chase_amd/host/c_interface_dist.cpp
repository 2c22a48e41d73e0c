// c_interface_dist.cpp — the DISTRIBUTED half of ChASE's application-facing C interface on the multi-GPU HIP Impl.
//
// Reference: interface/chase_c_interface.h:61-65,95-99,126-128,149,177-195 and interface/chase_c_interface.cpp:905-1290
// (ChASE_DIST<MatrixType>::Initialize: BlockCyclic / BlockBlock, Hermitian and pseudo-Hermitian), :1873-1921 (Solve),
// :3042-3230 (finalize, p?chase_, p?chase_get_eigenpairs_), :3543-3720 (p?chase_wrtHam_ / readHam_).
//
// The reference's init entry points take an MPI_Comm* and build the 2D grid from it.  Here the grid object is
// include/chase_hip_grid.h's chase_hip_grid (RCCL row / column communicators, one process per GPU), so every init has two
// forms:
//   p?chase_init*_hip_(..., chase_hip_grid* grid, int* init)   always in libchase_hip.so (this file): the caller made the
//       grid (chase_hip_grid_create_rccl with unique ids it distributed itself, or the host-callback transport);
//   p?chase_init*_(..., MPI_Comm* comm, int* init)             the reference's exact signatures, in libchase_hip_mpi.so
//       (chase_amd/host/c_interface_mpi.c, built when mpi.h is found): splits the communicator, broadcasts the RCCL ids over
//       MPI, creates the grid and calls the _hip_ form.
// Everything after init (p?chase_, p?chase_get_eigenpairs_, p?chase_finalize_, p?chase_readHam_ / wrtHam_) has no
// communicator in its signature and is the reference's name unchanged.
// Contract like the reference: H is the caller's HOST block (m x n local rows / columns, ldh), kept by pointer and copied to
// the device at the start of EVERY solve (pChASEGPU copies in initVecs, pchase_gpu.hpp:686 - an application may fill the
// block after init and change it between the solves of a sequence, examples/4_interface/4_c_dist_chase.c); V is the
// caller's host block of local rows (m x (nev+nex), ld = m; 2*(nev+nex) columns for pseudo-Hermitian problems) — read when
// mode == 'A', written after the solve (End() copies the eigenvectors back, pchase_gpu.hpp:1010-1018); irsrc / icsrc must
// be 0 (the reference's distribution functions assume it too, distMatrix.hpp:44-67 numroc with isrcproc = 0).
#include <atomic>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/chase_c_interface.h"
#include "../../include/chase_hip.h"
#include "../../include/chase_hip_grid.h"
#include "../../include/chase_hip_solver.h"

namespace chase_hip { int set_error(int code, const char* what); }

namespace {
struct DistSlot {
    chase_hip_solver* s = nullptr;
    chase_hip_ctx* ctx = nullptr;
    chase_hip_grid* grid = nullptr;
    bool own_grid = false;                     // grid + context were made by the MPI front end: finalize releases them
    int cplx = 0, pseudo = 0;
    std::size_t N = 0, nev = 0, nex = 0, ncol = 0, m = 0, n = 0, mb = 0, nb = 0;
    int nprow = 1, npcol = 1, myrow = 0, mycol = 0;
    void* dH = nullptr;
    void* H = nullptr;                         // the caller's host block, re-read at every solve
    int ldh = 0;
    void* V = nullptr;
    double* ritzv = nullptr;
    std::vector<double> own_v, own_ritzv;
    void clear()
    {
        if (s) { chase_hip_solver_destroy(s); s = nullptr; }
        if (dH && ctx) { chase_hip_free(ctx, dH); }
        dH = nullptr; H = nullptr; ldh = 0;
        if (own_grid) {
            if (grid) chase_hip_grid_destroy(grid);
            if (ctx) chase_hip_ctx_destroy(ctx);
        }
        grid = nullptr; ctx = nullptr; own_grid = false;
        own_v.clear(); own_v.shrink_to_fit(); own_ritzv.clear();
        V = nullptr; ritzv = nullptr;
    }
};
// One distributed solver per type like the reference's static members (chase_c_interface.cpp:905-1290): PROCESS-wide, so an
// application may call p?chase_init_ on one thread, initialise AGAIN from another without a finalize in between (the
// reference simply replaces its static solver) and call p?chase_ / get_eigenpairs / finalize from a third (OpenMP regions,
// a host language's worker threads, MPI_THREAD_MULTIPLE).  The one exception is EXPLICIT (round 5; rounds 3-4 inferred it
// from "a second thread initialises while a solver is alive", which captured exactly that legal re-initialisation): after
// chase_hip_cshim_thread_ranks(1) every thread that initialises is a RANK of a grid living in this process - one thread per
// GPU - and owns a slot of its own, which its later calls find by thread id.
std::atomic<bool> g_thread_ranks{false};
struct SlotTable {
    std::mutex mu;
    DistSlot global;
    std::map<std::thread::id, std::unique_ptr<DistSlot>> per_thread;
    DistSlot& for_init();
    // the calling thread ends (thread_local guard below): its own slot goes with it - thread ids are reused, and a stale
    // entry would capture the calls of whichever later thread happens to get the same id.  A rank thread that ends WITHOUT
    // p?chase_finalize_ (a failing test, an exception path) leaves a solver that points into a context and a grid it only
    // borrowed and that may be gone already: nothing borrowed is touched here - what the slot owns itself (the MPI front
    // end's grid + context) is released, a borrowed-handle solver is abandoned with a note (the advisor's finding).
    void drop_thread()
    {
        std::unique_ptr<DistSlot> mine;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = per_thread.find(std::this_thread::get_id());
            if (it == per_thread.end()) return;
            mine = std::move(it->second);
            per_thread.erase(it);
        }
        if (mine->s && !mine->own_grid) {
            std::fprintf(stderr, "chase_hip: a rank thread ended without p?chase_finalize_: its solver is abandoned (the context "
                                 "and grid it borrowed may be gone)\n");
            mine->s = nullptr; mine->dH = nullptr; mine->ctx = nullptr; mine->grid = nullptr;
        }
        mine->clear();
    }
    DistSlot& current()
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = per_thread.find(std::this_thread::get_id());
        return it != per_thread.end() ? *it->second : global;
    }
    void finalize()
    {
        DistSlot* mine = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = per_thread.find(std::this_thread::get_id());
            if (it != per_thread.end()) mine = it->second.get();   // stays this thread's (now empty) slot: what it sees from now on
        }
        (mine ? *mine : global).clear();
    }
};
SlotTable g_td, g_tz;
struct ThreadSlotGuard { ~ThreadSlotGuard() { g_td.drop_thread(); g_tz.drop_thread(); } };
thread_local ThreadSlotGuard g_thread_slot_guard;
DistSlot& SlotTable::for_init()
{
    std::lock_guard<std::mutex> lk(mu);
    const auto me = std::this_thread::get_id();
    auto it = per_thread.find(me);
    if (it != per_thread.end()) return *it->second;
    if (!g_thread_ranks.load()) return global;      // the reference's behaviour: one solver per type, replaced by a new init
    (void)&g_thread_slot_guard;                     // constructs this thread's guard: its destructor runs when the thread ends
    auto& up = per_thread[me];
    up.reset(new DistSlot());
    return *up;
}
#define g_pd (g_td.current())
#define g_pz (g_tz.current())
thread_local chase_hip_ctx* g_next_ctx = nullptr;           // set by chase_hip_cshim_use_ctx for the next init ON THIS THREAD
thread_local bool g_next_own = false;

// mbsize / nbsize == 0: block layout (block length rule of the reference, distMatrix.hpp:2000-2039)
void init_dist(DistSlot& sl, int cplx, int pseudo, int N, int nev, int nex, int mb, int nb, int m_in, int n_in, void* H,
               int ldh, void* V, double* ritzv, int irsrc, int icsrc, chase_hip_grid* grid, int* init)
{
    *init = 0;
    sl.clear();
    chase_hip_ctx* ctx = g_next_ctx;
    const bool own = g_next_own;
    g_next_ctx = nullptr; g_next_own = false;
    if (!grid || !ctx || !H || N <= 0 || nev <= 0 || nex < 0) {
        chase_hip::set_error(CHASE_HIP_EINVAL, "p?chase_init: NULL grid / context / matrix or bad sizes "
                                               "(call chase_hip_cshim_use_ctx before the _hip_ init)");
        if (own) { if (grid) chase_hip_grid_destroy(grid); if (ctx) chase_hip_ctx_destroy(ctx); }
        return;
    }
    sl.ctx = ctx; sl.grid = grid; sl.own_grid = own;
    if (irsrc != 0 || icsrc != 0) { chase_hip::set_error(CHASE_HIP_EINVAL, "p?chase_init_blockcyclic: irsrc / icsrc must be 0"); sl.clear(); return; }
    if (chase_hip_grid_info(grid, &sl.nprow, &sl.npcol, &sl.myrow, &sl.mycol)) { sl.clear(); return; }
    sl.cplx = cplx; sl.pseudo = pseudo;
    sl.N = (std::size_t)N; sl.nev = (std::size_t)nev; sl.nex = (std::size_t)nex;
    sl.ncol = (pseudo ? 2 : 1) * (std::size_t)(nev + nex);
    sl.mb = mb > 0 ? (std::size_t)mb : (std::size_t)chase_hip_block_len(N, sl.nprow);
    sl.nb = nb > 0 ? (std::size_t)nb : (std::size_t)chase_hip_block_len(N, sl.npcol);
    sl.m = (std::size_t)chase_hip_numroc(N, (long)sl.mb, sl.myrow, sl.nprow);
    sl.n = (std::size_t)chase_hip_numroc(N, (long)sl.nb, sl.mycol, sl.npcol);
    // the block entry points pass the local shape: it must be the one the layout implies (BlockBlockMatrix ctor,
    // distMatrix.hpp:1992-2052)
    if ((m_in >= 0 && (std::size_t)m_in != sl.m) || (n_in >= 0 && (std::size_t)n_in != sl.n) || (std::size_t)ldh < sl.m) {
        chase_hip::set_error(CHASE_HIP_EINVAL, "p?chase_init: local block shape does not match the layout (or ldh < m)");
        sl.clear();
        return;
    }
    const std::size_t es = cplx ? 16 : 8;
    sl.H = H; sl.ldh = ldh;
    if (chase_hip_malloc(ctx, &sl.dH, sl.m * sl.n * es)) { sl.clear(); return; }          // filled at every solve
    if (!V) {                                   // *_internal_: the interface owns the local V block and ritzv
        sl.own_v.assign(sl.m * sl.ncol * (cplx ? 2 : 1), 0.0);
        sl.own_ritzv.assign(sl.ncol, 0.0);
        V = sl.own_v.data(); ritzv = sl.own_ritzv.data();
    } else if (!ritzv) {
        sl.own_ritzv.assign(sl.ncol, 0.0);
        ritzv = sl.own_ritzv.data();
    }
    sl.V = V; sl.ritzv = ritzv;
    const int rc = pseudo ? chase_hip_psolver_create_pseudo(&sl.s, ctx, grid, cplx, sl.N, sl.nev, sl.nex, (size_t)(mb > 0 ? mb : 0),
                                                            (size_t)(nb > 0 ? nb : 0), sl.dH, sl.m, ritzv)
                          : chase_hip_psolver_create(&sl.s, ctx, grid, cplx, sl.N, sl.nev, sl.nex, (size_t)(mb > 0 ? mb : 0),
                                                     (size_t)(nb > 0 ? nb : 0), sl.dH, sl.m, ritzv);
    if (rc) { sl.clear(); return; }
    *init = 1;
}

// ChASE_DIST<MatrixType>::Solve (chase_c_interface.cpp:1873-1921)
void solve_dist(DistSlot& sl, int deg, double tol, char mode, char opt, char qr)
{
    if (!sl.s) return;
    chase_hip_solver_set(sl.s, "tol", tol);
    chase_hip_solver_set(sl.s, "deg", (double)deg);
    chase_hip_solver_set(sl.s, "opt", opt == 'S' ? 1.0 : 0.0);
    chase_hip_solver_set(sl.s, "approx", mode == 'A' ? 1.0 : 0.0);
    chase_hip_solver_set(sl.s, "cholqr", qr == 'C' ? 1.0 : 0.0);
    // the matrix as the caller's block holds it NOW (Hmat_->H2D() in initVecs, pchase_gpu.hpp:686)
    if (chase_hip_upload_matrix(sl.ctx, sl.cplx, (int)sl.m, (int)sl.n, sl.H, sl.ldh, sl.dH, (long)sl.m)) return;
    if (mode == 'A' && chase_hip_psolver_upload_v(sl.s, sl.V, sl.m)) return;       // approximate vectors from the caller
    if (chase_hip_solver_solve(sl.s, 0)) return;
    chase_hip_psolver_download_v(sl.s, sl.V, sl.m);                                 // End(): eigenvectors back to the host block
}
// copy_first_nev_results (chase_c_interface.cpp:3157-3230): local rows of the first nev eigenvectors + Ritz values
void get_pairs_dist(const DistSlot& sl, void* out, int ld, double* ritzv)
{
    if (!sl.s || !sl.V || !out || ld < (int)sl.m) return;
    const std::size_t es = sl.cplx ? 16 : 8;
    for (std::size_t j = 0; j < sl.nev; ++j)
        std::memcpy((char*)out + j * (std::size_t)ld * es, (const char*)sl.V + j * sl.m * es, sl.m * es);
    if (ritzv) std::memcpy(ritzv, sl.ritzv, sl.nev * sizeof(double));
}
int ham_io(DistSlot& sl, const char* filename, bool read)
{
    if (!sl.s || !filename) return chase_hip::set_error(CHASE_HIP_EINVAL, "p?chase_{read,wrt}Ham_: no initialised solver");
    if (read) {
        // the reference reads into the caller's host block (readFromBinaryFile on the CPU data, distMatrix.hpp:2425-2520);
        // here the shard goes through HBM and back so that the next solve's upload sees the loaded matrix
        int rc = chase_hip_load_matrix_shard(sl.ctx, filename, sl.cplx, (long)sl.N, (int)sl.m, (int)sl.n, (int)sl.mb, sl.nprow,
                                             sl.myrow, (int)sl.nb, sl.npcol, sl.mycol, sl.dH, (long)sl.m);
        if (rc) return rc;
        return chase_hip_download_matrix(sl.ctx, sl.cplx, (int)sl.m, (int)sl.n, sl.dH, (long)sl.m, sl.H, sl.ldh);
    }
    // the matrix the caller holds now (it may never have been solved with)
    if (int rc = chase_hip_upload_matrix(sl.ctx, sl.cplx, (int)sl.m, (int)sl.n, sl.H, sl.ldh, sl.dH, (long)sl.m)) return rc;
    return chase_hip_save_matrix_shard(sl.ctx, filename, sl.cplx, (long)sl.N, (int)sl.m, (int)sl.n, (int)sl.mb, sl.nprow,
                                       sl.myrow, (int)sl.nb, sl.npcol, sl.mycol, sl.dH, (long)sl.m);
}
} // namespace

extern "C" {

/* the device context the NEXT p?chase_init*_hip_ call uses (the grid was created on it); own != 0: that init's solver
 * takes ownership of grid and context and p?chase_finalize_ destroys them (what the MPI front end does) */
int chase_hip_cshim_use_ctx(chase_hip_ctx* ctx, int own)
{
    g_next_ctx = ctx;
    g_next_own = own != 0;
    return 0;
}
chase_hip_solver* chase_hip_cshim_dist_solver(int cplx) { return cplx ? g_pz.s : g_pd.s; }
/* on != 0: from now on every thread that calls a p?chase_init*_hip_ entry point is one RANK of a grid living in this process
 * (one thread per GPU) and gets a solver slot of its own, found again by thread id; 0 (default): the reference's process-wide
 * solver per type, which a new init from any thread replaces.  Returns the previous setting. */
int chase_hip_cshim_thread_ranks(int on) { return g_thread_ranks.exchange(on != 0) ? 1 : 0; }

/* ---- block layout (interface/chase_c_interface.h:126-149) ----------------------------------------------------------- */
void pdchase_init_hip_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, double* V, double* ritzv,
                       chase_hip_grid* grid, int* init)
{
    init_dist(g_td.for_init(), 0, 0, *N, *nev, *nex, 0, 0, *m, *n, H, *ldh, V, ritzv, 0, 0, grid, init);
}
void pdchase_init_internal_hip_(int* N, int* nev, int* nex, int* m, int* n, double* H, int* ldh, chase_hip_grid* grid, int* init)
{
    init_dist(g_td.for_init(), 0, 0, *N, *nev, *nex, 0, 0, *m, *n, H, *ldh, nullptr, nullptr, 0, 0, grid, init);
}
void pzchase_init_hip_(int* N, int* nev, int* nex, int* m, int* n, void* H, int* ldh, void* V, double* ritzv,
                       chase_hip_grid* grid, int* init)
{
    init_dist(g_tz.for_init(), 1, 0, *N, *nev, *nex, 0, 0, *m, *n, H, *ldh, V, ritzv, 0, 0, grid, init);
}
void pzchase_init_internal_hip_(int* N, int* nev, int* nex, int* m, int* n, void* H, int* ldh, chase_hip_grid* grid, int* init)
{
    init_dist(g_tz.for_init(), 1, 0, *N, *nev, *nex, 0, 0, *m, *n, H, *ldh, nullptr, nullptr, 0, 0, grid, init);
}
void pzchase_init_pseudo_hip_(int* N, int* nev, int* nex, int* m, int* n, void* H, int* ldh, void* V, double* ritzv,
                              chase_hip_grid* grid, int* init)
{
    init_dist(g_tz.for_init(), 1, 1, *N, *nev, *nex, 0, 0, *m, *n, H, *ldh, V, ritzv, 0, 0, grid, init);
}
void pzchase_init_pseudo_internal_hip_(int* N, int* nev, int* nex, int* m, int* n, void* H, int* ldh, chase_hip_grid* grid,
                                       int* init)
{
    init_dist(g_tz.for_init(), 1, 1, *N, *nev, *nex, 0, 0, *m, *n, H, *ldh, nullptr, nullptr, 0, 0, grid, init);
}
/* ---- block-cyclic layout (interface/chase_c_interface.h:61-124) ------------------------------------------------------ */
void pdchase_init_blockcyclic_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh, double* V,
                                   double* ritzv, int* irsrc, int* icsrc, chase_hip_grid* grid, int* init)
{
    init_dist(g_td.for_init(), 0, 0, *N, *nev, *nex, *mbsize, *nbsize, -1, -1, H, *ldh, V, ritzv, *irsrc, *icsrc, grid, init);
}
void pdchase_init_blockcyclic_internal_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, double* H, int* ldh,
                                            int* irsrc, int* icsrc, chase_hip_grid* grid, int* init)
{
    init_dist(g_td.for_init(), 0, 0, *N, *nev, *nex, *mbsize, *nbsize, -1, -1, H, *ldh, nullptr, nullptr, *irsrc, *icsrc, grid, init);
}
void pzchase_init_blockcyclic_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, void* H, int* ldh, void* V,
                                   double* ritzv, int* irsrc, int* icsrc, chase_hip_grid* grid, int* init)
{
    init_dist(g_tz.for_init(), 1, 0, *N, *nev, *nex, *mbsize, *nbsize, -1, -1, H, *ldh, V, ritzv, *irsrc, *icsrc, grid, init);
}
void pzchase_init_blockcyclic_internal_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, void* H, int* ldh,
                                            int* irsrc, int* icsrc, chase_hip_grid* grid, int* init)
{
    init_dist(g_tz.for_init(), 1, 0, *N, *nev, *nex, *mbsize, *nbsize, -1, -1, H, *ldh, nullptr, nullptr, *irsrc, *icsrc, grid, init);
}
void pzchase_init_pseudo_blockcyclic_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, void* H, int* ldh, void* V,
                                          double* ritzv, int* irsrc, int* icsrc, chase_hip_grid* grid, int* init)
{
    init_dist(g_tz.for_init(), 1, 1, *N, *nev, *nex, *mbsize, *nbsize, -1, -1, H, *ldh, V, ritzv, *irsrc, *icsrc, grid, init);
}
void pzchase_init_pseudo_blockcyclic_internal_hip_(int* N, int* nev, int* nex, int* mbsize, int* nbsize, void* H, int* ldh,
                                                   int* irsrc, int* icsrc, chase_hip_grid* grid, int* init)
{
    init_dist(g_tz.for_init(), 1, 1, *N, *nev, *nex, *mbsize, *nbsize, -1, -1, H, *ldh, nullptr, nullptr, *irsrc, *icsrc, grid, init);
}

/* ---- after init: the reference's names unchanged --------------------------------------------------------------------- */
void pdchase_(int* deg, double* tol, char* mode, char* opt, char* qr) { solve_dist(g_pd, *deg, *tol, *mode, *opt, *qr); }
void pzchase_(int* deg, double* tol, char* mode, char* opt, char* qr) { solve_dist(g_pz, *deg, *tol, *mode, *opt, *qr); }
void pdchase_get_eigenpairs_(double* LEigsV, int* ld, double* ritzv) { if (ld) get_pairs_dist(g_pd, LEigsV, *ld, ritzv); }
void pzchase_get_eigenpairs_(void* LEigsV, int* ld, double* ritzv) { if (ld) get_pairs_dist(g_pz, LEigsV, *ld, ritzv); }
void pdchase_finalize_(int* flag) { g_td.finalize(); if (flag) *flag = 0; }
void pzchase_finalize_(int* flag) { g_tz.finalize(); if (flag) *flag = 0; }
void pdchase_readHam_(const char* filename) { ham_io(g_pd, filename, true); }
void pzchase_readHam_(const char* filename) { ham_io(g_pz, filename, true); }
void pdchase_wrtHam_(const char* filename) { ham_io(g_pd, filename, false); }
void pzchase_wrtHam_(const char* filename) { ham_io(g_pz, filename, false); }

} // extern "C"
