// solver_capi.cpp — C entry points of the host solver: construct a ChaseHip<T> Impl, configure, solve, and drive the
// ChaseBase virtuals one by one (tests exercise every virtual through this surface, like the reference's
// tests/linalg unit tests do for its Impl kernels).  Declared in include/chase_hip_solver.h.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include "../../include/chase_hip.h"
#include "../../include/chase_hip_solver.h"
#include "../../include/chase_hip_grid.h"
#include "algorithm.hpp"
#include "chase_hip_impl.hpp"
#include "pchase_hip_impl.hpp"
#include "chase_hip_pseudo_impl.hpp"
#include "pchase_hip_pseudo_impl.hpp"
#include "tape.hpp"

namespace chase_hip { int set_error(int code, const char* what); }
using namespace chase_amd;

using zc = std::complex<double>;
struct chase_hip_solver {
    int cplx = 0, pseudo = 0;
    std::unique_ptr<ChaseBase<double>> d;           // ChaseHip<double> or pChaseHip<double>
    std::unique_ptr<ChaseBase<zc>> z;
    HipImplExtras* ex = nullptr;                    // same object, Impl-specific extras
    pChaseHip<double>* pd = nullptr;                // set for the distributed Impl only
    pChaseHip<zc>* pz = nullptr;
    SolveStats stats;
    CallTrace trace;
    std::string trace_text;
    int tape_mode = 0;                              // 0 off, 1 record, 2 replay (tape.hpp)
    ScalarTape tape;
};

namespace {
template <class F>
int guarded(const char* where, F&& f)
{
    try {
        f();
        return 0;
    } catch (const HipStatusError& e) {
        chase_hip::set_error(e.code, e.what());
        return e.code;
    } catch (const std::invalid_argument& e) {
        return chase_hip::set_error(CHASE_HIP_EINVAL, (std::string(where) + ": " + e.what()).c_str());
    } catch (const std::exception& e) {
        return chase_hip::set_error(CHASE_HIP_EINVAL, (std::string(where) + ": " + e.what()).c_str());
    }
}
#define DISPATCH(s, expr)                                                                                              \
    do {                                                                                                               \
        if ((s)->cplx) { auto& k = *(s)->z; expr; } else { auto& k = *(s)->d; expr; }                                  \
    } while (0)
} // namespace

extern "C" {

int chase_hip_solver_create(chase_hip_solver** out, chase_hip_ctx* ctx, int cplx, size_t N, size_t nev, size_t nex,
                            void* H, size_t ldh, void* V, size_t ldv, double* ritzv, int h_on_device)
{
    if (!out || !ctx || !H || !V || !ritzv) return chase_hip::set_error(CHASE_HIP_EINVAL, "solver_create: NULL argument");
    auto s = std::make_unique<chase_hip_solver>();
    s->cplx = cplx ? 1 : 0;
    chase_hip_host_lapack_warmup();          // not fatal here: a missing provider is reported by the first LAPACK call
    int rc = guarded("solver_create", [&] {
        if (cplx) {
            auto* p = new ChaseHip<zc>(ctx, N, nev, nex, (zc*)H, ldh, (zc*)V, ldv, ritzv, h_on_device != 0);
            s->z.reset(p); s->ex = p;
        } else {
            auto* p = new ChaseHip<double>(ctx, N, nev, nex, (double*)H, ldh, (double*)V, ldv, ritzv, h_on_device != 0);
            s->d.reset(p); s->ex = p;
        }
    });
    if (rc) return rc;
    *out = s.release();
    return 0;
}

/* pseudo-Hermitian (BSE) sequential Impl: V is N x 2*(nev+nex), ritzv has 2*(nev+nex) entries */
int chase_hip_solver_create_pseudo(chase_hip_solver** out, chase_hip_ctx* ctx, int cplx, size_t N, size_t nev, size_t nex,
                                   void* H, size_t ldh, void* V, size_t ldv, double* ritzv, int h_on_device)
{
    if (!out || !ctx || !H || !V || !ritzv) return chase_hip::set_error(CHASE_HIP_EINVAL, "solver_create_pseudo: NULL argument");
    auto s = std::make_unique<chase_hip_solver>();
    s->cplx = cplx ? 1 : 0;
    s->pseudo = 1;
    chase_hip_host_lapack_warmup();          // not fatal here: a missing provider is reported by the first LAPACK call
    int rc = guarded("solver_create_pseudo", [&] {
        if (cplx) {
            auto* p = new ChaseHipPseudo<zc>(ctx, N, nev, nex, (zc*)H, ldh, (zc*)V, ldv, ritzv, h_on_device != 0);
            s->z.reset(p); s->ex = p;
        } else {
            auto* p = new ChaseHipPseudo<double>(ctx, N, nev, nex, (double*)H, ldh, (double*)V, ldv, ritzv, h_on_device != 0);
            s->d.reset(p); s->ex = p;
        }
    });
    if (rc) return rc;
    *out = s.release();
    return 0;
}

int chase_hip_psolver_create(chase_hip_solver** out, chase_hip_ctx* ctx, chase_hip_grid* grid, int cplx, size_t N,
                             size_t nev, size_t nex, size_t mb, size_t nb, void* H_loc_dev, size_t ldh, double* ritzv)
{
    if (!out || !ctx || !grid || !H_loc_dev || !ritzv)
        return chase_hip::set_error(CHASE_HIP_EINVAL, "psolver_create: NULL argument");
    auto s = std::make_unique<chase_hip_solver>();
    s->cplx = cplx ? 1 : 0;
    chase_hip_host_lapack_warmup();          // not fatal here: a missing provider is reported by the first LAPACK call
    int rc = guarded("psolver_create", [&] {
        if (cplx) {
            auto* p = new pChaseHip<zc>(ctx, grid, N, nev, nex, mb, nb, (zc*)H_loc_dev, ldh, ritzv);
            s->z.reset(p); s->ex = p; s->pz = p;
        } else {
            auto* p = new pChaseHip<double>(ctx, grid, N, nev, nex, mb, nb, (double*)H_loc_dev, ldh, ritzv);
            s->d.reset(p); s->ex = p; s->pd = p;
        }
    });
    if (rc) return rc;
    *out = s.release();
    return 0;
}

/* distributed pseudo-Hermitian (BSE) Impl: 2*(nev+nex) vector columns, ritzv has 2*(nev+nex) entries */
int chase_hip_psolver_create_pseudo(chase_hip_solver** out, chase_hip_ctx* ctx, chase_hip_grid* grid, int cplx, size_t N,
                                    size_t nev, size_t nex, size_t mb, size_t nb, void* H_loc_dev, size_t ldh,
                                    double* ritzv)
{
    if (!out || !ctx || !grid || !H_loc_dev || !ritzv)
        return chase_hip::set_error(CHASE_HIP_EINVAL, "psolver_create_pseudo: NULL argument");
    auto s = std::make_unique<chase_hip_solver>();
    s->cplx = cplx ? 1 : 0;
    s->pseudo = 1;
    chase_hip_host_lapack_warmup();          // not fatal here: a missing provider is reported by the first LAPACK call
    int rc = guarded("psolver_create_pseudo", [&] {
        if (cplx) {
            auto* p = new pChaseHipPseudo<zc>(ctx, grid, N, nev, nex, mb, nb, (zc*)H_loc_dev, ldh, ritzv);
            s->z.reset(p); s->ex = p; s->pz = p;
        } else {
            auto* p = new pChaseHipPseudo<double>(ctx, grid, N, nev, nex, mb, nb, (double*)H_loc_dev, ldh, ritzv);
            s->d.reset(p); s->ex = p; s->pd = p;
        }
    });
    if (rc) return rc;
    *out = s.release();
    return 0;
}

/* local shape of the distributed solver's blocks: rows of V (== rows of H_loc) and columns of H_loc */
int chase_hip_psolver_local_shape(chase_hip_solver* s, size_t* m_loc, size_t* n_loc)
{
    if (!s || (!s->pd && !s->pz)) return chase_hip::set_error(CHASE_HIP_EINVAL, "not a distributed solver");
    if (m_loc) *m_loc = s->ex->local_rows();
    if (n_loc) *n_loc = s->pz ? s->pz->local_cols_h() : s->pd->local_cols_h();
    return 0;
}
int chase_hip_psolver_upload_v(chase_hip_solver* s, const void* host, size_t ldv)
{
    if (!s || (!s->pd && !s->pz)) return chase_hip::set_error(CHASE_HIP_EINVAL, "not a distributed solver");
    return guarded("upload_v", [&] { if (s->pz) s->pz->upload_local_V((const zc*)host, ldv); else s->pd->upload_local_V((const double*)host, ldv); });
}
int chase_hip_psolver_download_v(chase_hip_solver* s, void* host, size_t ldv)
{
    if (!s || (!s->pd && !s->pz)) return chase_hip::set_error(CHASE_HIP_EINVAL, "not a distributed solver");
    return guarded("download_v", [&] { if (s->pz) s->pz->download_local_V((zc*)host, ldv); else s->pd->download_local_V((double*)host, ldv); });
}
int chase_hip_psolver_set_pipeline(chase_hip_solver* s, int on)
{
    if (!s || (!s->pd && !s->pz)) return chase_hip::set_error(CHASE_HIP_EINVAL, "not a distributed solver");
    if (s->pz) s->pz->set_pipeline(on != 0); else s->pd->set_pipeline(on != 0);
    return 0;
}

int chase_hip_solver_destroy(chase_hip_solver* s)
{
    delete s;
    return 0;
}

int chase_hip_solver_set(chase_hip_solver* s, const char* key, double v)
{
    if (!s || !key) return chase_hip::set_error(CHASE_HIP_EINVAL, "solver_set: NULL argument");
    const std::string name(key);
    int rc = 0;
    const int grc = guarded("solver_set", [&] { DISPATCH(s, {
        auto& c = k.GetConfig();
        if (name == "tol") c.SetTol(v);
        else if (name == "deg") c.SetDeg((size_t)v);
        else if (name == "maxdeg") c.SetMaxDeg((size_t)v);
        else if (name == "degextra") c.SetDegExtra((size_t)v);
        else if (name == "maxiter") c.SetMaxIter((size_t)v);
        else if (name == "lanczositer") c.SetLanczosIter((size_t)v);
        else if (name == "numlanczos") c.SetNumLanczos((size_t)v);
        else if (name == "opt") c.SetOpt(v != 0);
        else if (name == "approx") c.SetApprox(v != 0);
        else if (name == "cholqr") c.SetCholQR(v != 0);
        else if (name == "decayingrate") c.SetDecayingRate((float)v);
        else if (name == "clusteraware") c.SetClusterAwareDegrees(v != 0);
        else if (name == "upperbscale") c.SetUpperbScaleRate((float)v);
        else if (name == "device_rng") s->ex->set_device_rng(v != 0);
        else if (name == "panel_cols" && (s->pd || s->pz)) { if (s->pz) s->pz->set_panel_cols((size_t)v); else s->pd->set_panel_cols((size_t)v); }
        else if (name == "panel_rounds" && (s->pd || s->pz)) { if (s->pz) s->pz->set_panel_rounds((int)v); else s->pd->set_panel_rounds((int)v); }
        else if (name == "pipeline" && (s->pd || s->pz)) { if (s->pz) s->pz->set_pipeline(v != 0); else s->pd->set_pipeline(v != 0); }
        else if (name == "reset_counters") s->ex->reset_counters();
        else rc = chase_hip::set_error(CHASE_HIP_EINVAL, "solver_set: unknown key");
    }); });
    return grc ? grc : rc;
}

int chase_hip_solver_get(chase_hip_solver* s, const char* key, double* out)
{
    if (!s || !key || !out) return chase_hip::set_error(CHASE_HIP_EINVAL, "solver_get: NULL argument");
    const std::string name(key);
    int rc = 0;
    DISPATCH(s, {
        auto& c = k.GetConfig();
        if (name == "tol") *out = c.GetTol();
        else if (name == "deg") *out = (double)c.GetDeg();
        else if (name == "maxdeg") *out = (double)c.GetMaxDeg();
        else if (name == "degextra") *out = (double)c.GetDegExtra();
        else if (name == "maxiter") *out = (double)c.GetMaxIter();
        else if (name == "lanczositer") *out = (double)c.GetLanczosIter();
        else if (name == "numlanczos") *out = (double)c.GetNumLanczos();
        else if (name == "opt") *out = c.DoOptimization();
        else if (name == "approx") *out = c.UseApprox();
        else if (name == "cholqr") *out = c.DoCholQR();
        else if (name == "decayingrate") *out = c.GetDecayingRate();
        else if (name == "clusteraware") *out = c.UseClusterAwareDegrees();
        else if (name == "upperbscale") *out = c.GetUpperbScaleRate();
        else if (name == "locked") *out = (double)s->ex->locked();
        else if (name == "qr_ortho_check" && (s->pd || s->pz)) *out = s->pz ? s->pz->last_ortho_check() : s->pd->last_ortho_check();
        else if (name == "panel_cols" && (s->pd || s->pz)) *out = (double)(s->pz ? s->pz->panel_cols() : s->pd->panel_cols());
        else if (name == "panel_rounds" && (s->pd || s->pz)) *out = (double)(s->pz ? s->pz->panel_rounds() : s->pd->panel_rounds());
        else if (name == "pipeline" && (s->pd || s->pz)) *out = (double)(s->pz ? s->pz->pipeline() : s->pd->pipeline());
        else if (name == "rr_disagreements" && (s->pd || s->pz)) *out = (double)(s->pz ? s->pz->rr_disagreements() : s->pd->rr_disagreements());
        else if (name == "qr_variant") *out = (double)s->ex->last_qr_variant();
        else if (name == "filter_ms") *out = s->ex->filter_ms();
        else if (name == "hemm_calls") *out = (double)s->ex->hemm_calls();
        else if (name == "hemm_reused_vecs") *out = (double)s->ex->hemm_reused_vecs();
        else if (name == "resd_rechecked") *out = (double)s->ex->resd_rechecked();
        else if (name == "tape_qr_mismatches") *out = (double)s->tape.qr_variant_mismatches;   // of the last replay
        else if (name == "tape_qr_retries") *out = (double)s->ex->forced_qr_retries();   // shifted re-factorisations, replay
        else if (name == "tape_tolerated") *out = (double)s->ex->replay_tolerated();    // pseudo-Hermitian replay: tolerated cores
        else if (name == "tape_position") *out = (double)s->tape.pos;
        else if (name == "tape_size") *out = (double)s->tape.data.size();
        else if (name == "iterations") *out = (double)s->stats.iterations;          // of the last solve
        else if (name == "filtered_vecs") *out = (double)s->stats.filtered_vecs;
        else rc = chase_hip::set_error(CHASE_HIP_EINVAL, "solver_get: unknown key");
    });
    return rc;
}

int chase_hip_solver_set_iteration_hook(chase_hip_solver* s, chase_hip_iteration_fn fn, void* user)
{
    if (!s) return chase_hip::set_error(CHASE_HIP_EINVAL, "set_iteration_hook: NULL solver");
    s->trace.iter_hook = fn;
    s->trace.iter_user = user;
    return 0;
}

int chase_hip_solver_solve(chase_hip_solver* s, int record_trace)
{
    if (!s) return chase_hip::set_error(CHASE_HIP_EINVAL, "solver_solve: NULL solver");
    s->stats = SolveStats();
    s->trace.lines.clear();
    s->trace.enabled = record_trace != 0;
    return guarded("solve", [&] {
        // chase::Solve / chase::Solve_pseudo (algorithm/algorithm.hpp:345-364)
        if (s->pseudo && s->tape_mode) {
            if (s->cplx) {
                TapeKernel<zc> tk(s->z.get(), s->ex, &s->tape, (TapeKernel<zc>::Mode)s->tape_mode);
                Algorithm<zc, ChaseBase<zc>>::solve_pseudo(&tk, &s->stats, &s->trace);
            } else {
                TapeKernel<double> tk(s->d.get(), s->ex, &s->tape, (TapeKernel<double>::Mode)s->tape_mode);
                Algorithm<double, ChaseBase<double>>::solve_pseudo(&tk, &s->stats, &s->trace);
            }
        } else if (s->pseudo) {
            if (s->cplx) Algorithm<zc, ChaseBase<zc>>::solve_pseudo(s->z.get(), &s->stats, &s->trace);
            else Algorithm<double, ChaseBase<double>>::solve_pseudo(s->d.get(), &s->stats, &s->trace);
        } else if (s->tape_mode) {
            // the unmodified driver on the taping decorator: record the kernel's host outputs, or replay recorded ones
            if (s->cplx) {
                TapeKernel<zc> tk(s->z.get(), s->ex, &s->tape, (TapeKernel<zc>::Mode)s->tape_mode);
                Algorithm<zc, ChaseBase<zc>>::solve(&tk, &s->stats, &s->trace);
            } else {
                TapeKernel<double> tk(s->d.get(), s->ex, &s->tape, (TapeKernel<double>::Mode)s->tape_mode);
                Algorithm<double, ChaseBase<double>>::solve(&tk, &s->stats, &s->trace);
            }
        } else {
            if (s->cplx) Algorithm<zc, ChaseBase<zc>>::solve(s->z.get(), &s->stats, &s->trace);
            else Algorithm<double, ChaseBase<double>>::solve(s->d.get(), &s->stats, &s->trace);
        }
    });
}

/* Scalar tape of chase_hip_solver_solve (chase_amd/host/tape.hpp): mode 1 = the next solves RECORD everything the kernel
 * tells the driver (Ritz values, residuals, Lanczos outputs, the QR variant taken), mode 2 = the next solves REPLAY the loaded
 * tape: the kernel does all of its device work, the driver sees the recorded numbers and therefore issues the recorded call
 * sequence; 0 = off.  Both drivers (chase::Solve, chase::Solve_pseudo). */
int chase_hip_solver_tape_mode(chase_hip_solver* s, int mode)
{
    if (!s || mode < 0 || mode > 2) return chase_hip::set_error(CHASE_HIP_EINVAL, "tape_mode: 0, 1 or 2");
    s->tape_mode = mode;
    return 0;
}
int chase_hip_solver_tape_data(chase_hip_solver* s, const double** data, size_t* count)
{
    if (!s || !data || !count) return chase_hip::set_error(CHASE_HIP_EINVAL, "tape_data: NULL argument");
    *data = s->tape.data.data();
    *count = s->tape.data.size();
    return 0;
}
int chase_hip_solver_tape_load(chase_hip_solver* s, const double* data, size_t count)
{
    if (!s || (!data && count)) return chase_hip::set_error(CHASE_HIP_EINVAL, "tape_load: NULL argument");
    s->tape.data.assign(data, data + count);
    s->tape.rewind();
    return 0;
}

/* Algorithm<T>::lanczos_for_H2 (algorithm/algorithm.inc:1217-1373) on a pseudo-Hermitian solver: DoS estimates of the
 * H^2 spectrum into the solver's ritzv[0 .. nev+nex), upper bound and the number of extracted Ritz directions */
int chase_hip_solver_lanczos_for_h2(chase_hip_solver* s, int numvec, int m, double* upperb, size_t* idx)
{
    if (!s || !upperb || !idx) return chase_hip::set_error(CHASE_HIP_EINVAL, "lanczos_for_h2: NULL argument");
    if (!s->pseudo) return chase_hip::set_error(CHASE_HIP_EINVAL, "lanczos_for_h2: not a pseudo-Hermitian solver");
    return guarded("lanczos_for_H2", [&] {
        if (s->cplx) {
            auto* k = s->z.get();
            *idx = Algorithm<zc, ChaseBase<zc>>::lanczos_for_H2(k, (int)k->GetN(), numvec, m, (int)(k->GetNev() + k->GetNex()),
                                                                 upperb, k->GetRitzv());
        } else {
            auto* k = s->d.get();
            *idx = Algorithm<double, ChaseBase<double>>::lanczos_for_H2(k, (int)k->GetN(), numvec, m,
                                                                        (int)(k->GetNev() + k->GetNex()), upperb, k->GetRitzv());
        }
    });
}

int chase_hip_solver_stats(chase_hip_solver* s, chase_hip_stats* o)
{
    if (!s || !o) return chase_hip::set_error(CHASE_HIP_EINVAL, "solver_stats: NULL argument");
    const SolveStats& t = s->stats;
    o->iterations = t.iterations; o->filtered_vecs = t.filtered_vecs; o->lanczos_vecs = t.lanczos_vecs;
    o->locked = t.locked;
    o->t_all = t.t_all; o->t_init = t.t_init; o->t_lanczos = t.t_lanczos; o->t_filter = t.t_filter; o->t_qr = t.t_qr;
    o->t_rr = t.t_rr; o->t_resid = t.t_resid;
    o->lowerb = t.lowerb; o->upperb = t.upperb; o->lambda = t.lambda;
    double fm = 0;
    chase_hip_solver_get(s, "filter_ms", &fm);
    o->filter_ms_device = fm;
    return 0;
}

const double* chase_hip_solver_resid(chase_hip_solver* s)
{
    if (!s) return nullptr;
    const double* r = nullptr;
    DISPATCH(s, r = k.GetResid());
    return r;
}

const char* chase_hip_solver_trace(chase_hip_solver* s)
{
    if (!s) return "";
    s->trace_text.clear();
    for (const auto& l : s->trace.lines) { s->trace_text += l; s->trace_text += '\n'; }
    return s->trace_text.c_str();
}

/* ---- the ChaseBase virtuals, one entry point each (algorithm/interface.hpp:60-433) ---------------------------- */
int chase_hip_op_start(chase_hip_solver* s) { return guarded("Start", [&] { DISPATCH(s, k.Start()); }); }
int chase_hip_op_end(chase_hip_solver* s) { return guarded("End", [&] { DISPATCH(s, k.End()); }); }
int chase_hip_op_reinit_columns(chase_hip_solver* s, size_t fixednev, const size_t* col_indices, size_t n_indices)
{
    if (!s || (!col_indices && n_indices)) return chase_hip::set_error(CHASE_HIP_EINVAL, "ReinitColumns: NULL argument");
    return guarded("ReinitColumns", [&] { DISPATCH(s, k.ReinitColumns(fixednev, col_indices, n_indices)); });
}
int chase_hip_op_initvecs(chase_hip_solver* s, int random)
{
    return guarded("initVecs", [&] { DISPATCH(s, k.initVecs(random != 0)); });
}
int chase_hip_op_shift(chase_hip_solver* s, double c, int isunshift)
{
    return guarded("Shift", [&] {
        if (s->cplx) s->z->Shift(std::complex<double>(c, 0), isunshift != 0);
        else s->d->Shift(c, isunshift != 0);
    });
}
int chase_hip_op_hemm(chase_hip_solver* s, size_t block, const double* alpha, const double* beta, size_t offset_left,
                      size_t offset_right)
{
    return guarded("HEMM", [&] {
        if (s->cplx)
            s->z->HEMM(block, std::complex<double>(alpha[0], alpha[1]), std::complex<double>(beta[0], beta[1]),
                       offset_left, offset_right);
        else s->d->HEMM(block, alpha[0], beta[0], offset_left, offset_right);
    });
}
int chase_hip_op_hemm_h2(chase_hip_solver* s, size_t block, const double* alpha, const double* beta, const double* gamma,
                         size_t offset_left, size_t offset_right)
{
    return guarded("HEMM_H2", [&] {
        if (s->cplx)
            s->z->HEMM_H2(block, zc(alpha[0], alpha[1]), zc(beta[0], beta[1]), zc(gamma[0], gamma[1]), offset_left, offset_right);
        else s->d->HEMM_H2(block, alpha[0], beta[0], gamma[0], offset_left, offset_right);
    });
}
int chase_hip_op_kconj(chase_hip_solver* s, size_t block)
{
    return guarded("ApplyKconjugate", [&] { DISPATCH(s, k.ApplyKconjugate(block)); });
}
int chase_hip_op_qr(chase_hip_solver* s, size_t fixednev, double cond)
{
    return guarded("QR", [&] { DISPATCH(s, k.QR(fixednev, cond)); });
}
int chase_hip_op_rr(chase_hip_solver* s, double* ritzv, size_t block)
{
    return guarded("RR", [&] { DISPATCH(s, k.RR(ritzv, block)); });
}
int chase_hip_op_resd(chase_hip_solver* s, double* ritzv, double* resd, size_t fixednev)
{
    return guarded("Resd", [&] { DISPATCH(s, k.Resd(ritzv, resd, fixednev)); });
}
int chase_hip_op_swap(chase_hip_solver* s, size_t i, size_t j) { return guarded("Swap", [&] { DISPATCH(s, k.Swap(i, j)); }); }
int chase_hip_op_lock(chase_hip_solver* s, size_t n) { return guarded("Lock", [&] { DISPATCH(s, k.Lock(n)); }); }
int chase_hip_op_lanczos(chase_hip_solver* s, size_t M, size_t numvec, double* upperb, double* ritzv, double* Tau,
                         double* ritzV)
{
    return guarded("Lanczos", [&] {
        if (numvec == 0) DISPATCH(s, k.Lanczos(M, upperb));
        else DISPATCH(s, k.Lanczos(M, numvec, upperb, ritzv, Tau, ritzV));
    });
}
int chase_hip_op_lanczos_dos(chase_hip_solver* s, size_t idx, size_t m, void* ritzVc)
{
    return guarded("LanczosDos", [&] {
        if (s->cplx) s->z->LanczosDos(idx, m, (std::complex<double>*)ritzVc);
        else s->d->LanczosDos(idx, m, (double*)ritzVc);
    });
}
int chase_hip_op_sym_or_herm(chase_hip_solver* s, char uplo)
{
    return guarded("symOrHermMatrix", [&] { DISPATCH(s, k.symOrHermMatrix(uplo)); });
}
int chase_hip_op_check_symmetry(chase_hip_solver* s, int* is_sym)
{
    return guarded("checkSymmetryEasy", [&] { DISPATCH(s, *is_sym = k.checkSymmetryEasy() ? 1 : 0); });
}
/* copies the current device V1 (N x nevex) to a host buffer without ending the solve (tests) */
int chase_hip_solver_recompute_residuals(chase_hip_solver* s, size_t ncols, const double* lambda, double* resid)
{
    if (!s || !lambda || !resid) return chase_hip::set_error(CHASE_HIP_EINVAL, "recompute_residuals: NULL argument");
    return guarded("recompute_residuals", [&] { s->ex->recompute_residuals(ncols, lambda, resid); });
}

/* 64-bit content hash of the first ncols columns of the (local) vector block as it sits in HBM */
int chase_hip_solver_hash_v(chase_hip_solver* s, chase_hip_ctx* ctx, size_t ncols, unsigned long long* hash)
{
    if (!s || !ctx || !hash) return chase_hip::set_error(CHASE_HIP_EINVAL, "hash_v: NULL argument");
    return guarded("hash_v", [&] {
        DISPATCH(s, {
            if (ncols > k.GetRitzvBlockSize()) throw std::invalid_argument("more columns than the Impl holds");
            hip_ok(chase_hip_hash64(ctx, s->cplx, (int)s->ex->local_rows(), (int)ncols, s->ex->device_V1(), (long)s->ex->local_rows(), hash), "hash64");
        });
    });
}

int chase_hip_solver_peek_v(chase_hip_solver* s, chase_hip_ctx* ctx, void* host, size_t ldh)
{
    return guarded("peek_v", [&] {
        DISPATCH(s, hip_ok(chase_hip_download_matrix(ctx, s->cplx, (int)s->ex->local_rows(), (int)k.GetRitzvBlockSize(),
                                                     s->ex->device_V1(), (long)s->ex->local_rows(), host, (long)ldh),
                           "download"));
    });
}

} // extern "C"
