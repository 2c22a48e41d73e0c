// chase_hip_impl.hpp — ChaseHip<T, BaseT>: single-MI355X implementation of the ChaseBase<T> operator surface.
//
// Mirrors the semantics of the reference's sequential Impls virtual by virtual
//   Impl/chase_cpu/chase_cpu.hpp:294-841 (ChASECPU)   and   Impl/chase_gpu/chase_gpu.hpp:400-1018 (ChASEGPU)
// but every matrix operation is a call through the C ABI of include/chase_hip.h into hand-written gfx950 kernels.
// H, V1 and ritzv are caller-owned (host pointers, or device pointers when flagged); V2, A, scratch are owned here.
//
// Differences from the reference that are deliberate (results unchanged up to rounding):
//   * Swap() is deferred: swaps only update a host permutation; the next operation that reads the vectors applies
//     the whole permutation in two launches (calc_degrees can issue O(n^2) swaps, algorithm.inc:181-190).
//   * Lanczos keeps all scalars on the device (one host sync per Lanczos call instead of several per step).
//
// BaseT lets the same class derive from chase::ChaseBase<T> inside a ChASE checkout.
#pragma once
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/chase_hip.h"
#include "impl_extras.hpp"
#include "interface.hpp"
#include "output_override.hpp"
#include "roctx.hpp"

namespace chase_amd {

template <class T> struct is_cplx : std::false_type {};
template <class R> struct is_cplx<std::complex<R>> : std::true_type {};

struct HipStatusError : std::runtime_error {
    int code;
    HipStatusError(int c, const std::string& where)
        : std::runtime_error(where + " failed (" + std::to_string(c) + "): " + chase_hip_last_error()), code(c) {}
};
inline void hip_ok(int rc, const char* where) { if (rc < 0) throw HipStatusError(rc, where); }

template <class T, class BaseT = ChaseBase<T>, class ConfigT = ChaseConfig<T>>
class ChaseHip : public WithOutput<BaseT>, public HipImplExtras {
public:
    using R = Base<T>;
    static constexpr int CP = is_cplx<T>::value ? 1 : 0;

    // H: N x N column-major (ldh), V1: N x (nev+nex) (ldv), ritzv: nev+nex reals.  h_on_device: H already lives in HBM
    // (it is then used in place, shifted and unshifted by the filter exactly like the reference does to its copy).
    ChaseHip(chase_hip_ctx* ctx, std::size_t N, std::size_t nev, std::size_t nex, T* H, std::size_t ldh, T* V1,
             std::size_t ldv, R* ritzv, bool h_on_device = false)
        : ctx_(ctx), N_(N), nev_(nev), nex_(nex), nevex_(nev + nex), H_(H), ldh_(ldh), V1_(V1), ldv_(ldv),
          ritzv_(ritzv), h_on_device_(h_on_device), config_(N, nev, nex), resid_(nev + nex, 0), perm_(nev + nex)
    {
        if (!ctx) throw std::invalid_argument("ChaseHip: null context");
        if (N == 0 || nevex_ == 0 || nevex_ > N) throw std::invalid_argument("ChaseHip: need 0 < nev+nex <= N");
        if (ldh < N || ldv < N) throw std::invalid_argument("ChaseHip: leading dimension smaller than N");
        for (std::size_t i = 0; i < nevex_; ++i) perm_[i] = (int)i;
        if (h_on_device_) { dH_ = H; ldd_h_ = ldh; }
        else { alloc((void**)&dH_, N_ * N_ * sizeof(T)); ldd_h_ = N_; own_h_ = true; }
        alloc((void**)&dV1_, N_ * nevex_ * sizeof(T));
        alloc((void**)&dV2_, N_ * nevex_ * sizeof(T));
        alloc((void**)&dA_, nevex_ * nevex_ * sizeof(T));
        alloc((void**)&dScal_, 4096);
    }
    ~ChaseHip() override
    {
        for (void* p : owned_) chase_hip_free(ctx_, p);
    }

    // ---- trivial getters ---------------------------------------------------------------------------------------
    std::size_t GetN() const override { return N_; }
    std::size_t GetNev() override { return nev_; }
    std::size_t GetNex() override { return nex_; }
    std::size_t GetLanczosIter() override { return lanczosIter_; }
    std::size_t GetNumLanczos() override { return numLanczos_; }
    std::size_t GetRitzvBlockSize() const override { return nevex_; }
    R* GetRitzv() override { return ritzv_; }
    R* GetResid() override { return resid_.data(); }
    ConfigT& GetConfig() override { return config_; }
    int get_nprocs() override { return 1; }
    int get_rank() override { return 0; }
    bool isSym() override { return true; }
    bool isPseudoHerm() override { return false; }
    bool checkPseudoHermicityEasy() override { return false; }
    void Sort(R*, R*, R*) override {}
    void ApplyKconjugate(std::size_t) override {}
    void HEMM_H2(std::size_t, T, T, T, std::size_t, std::size_t = 0) override
    {
        throw std::logic_error("ChaseHip: HEMM_H2 belongs to the pseudo-Hermitian Impl");
    }
    void set_early_locked_residuals(std::vector<R> r) override { early_ = std::move(r); }
    const std::vector<R>& early_locked_residuals() const { return early_; }
    std::size_t locked() const override { return locked_; }
    void* device_V1() override { flush_swaps(); return dV1_; }
    std::size_t local_rows() const override { return N_; }
    T* device_V2() { return dV2_; }
    T* device_H() { return dH_; }
    double filter_ms() const override { return filter_ms_; }

    // randomized Hermiticity check: ||H v - H^H v|| small  (reference: cpu::checkSymmetryEasy, symOrHerm.hpp)
    bool checkSymmetryEasy() override
    {
        CHASE_PHASE(ctx_, "checkSymmetryEasy");
        if (!h_resident_) upload_H();
        std::vector<T> v(N_);
        std::mt19937 gen(1337);
        std::normal_distribution<> d;
        for (auto& x : v) x = rnd(d, gen);
        T* u = dV2_;                     // scratch: three columns of V2
        hip_ok(chase_hip_upload_matrix(ctx_, CP, (int)N_, 1, v.data(), (long)N_, u, (long)N_), "upload");
        gemm('N', N_, 1, N_, T(1), dH_, ldd_h_, u, N_, T(0), u + N_, N_);
        gemm('C', N_, 1, N_, T(1), dH_, ldd_h_, u, N_, T(0), u + 2 * N_, N_);
        std::vector<T> a(N_), b(N_);
        hip_ok(chase_hip_download_matrix(ctx_, CP, (int)N_, 1, u + N_, (long)N_, a.data(), (long)N_), "download");
        hip_ok(chase_hip_download_matrix(ctx_, CP, (int)N_, 1, u + 2 * N_, (long)N_, b.data(), (long)N_), "download");
        double diff = 0, nrm = 0;
        for (std::size_t i = 0; i < N_; ++i) { diff += std::norm(a[i] - b[i]); nrm += std::norm(a[i]); }
        return std::sqrt(diff) <= 1e-10 * std::max(1.0, std::sqrt(nrm));
    }

    // complete the other triangle from `uplo` on the caller's host copy (reference: cpu::symOrHermMatrix)
    void symOrHermMatrix(char uplo) override
    {
        if (h_on_device_) {             // the caller's matrix lives in HBM: completed in place there (round 5)
            hip_ok(chase_hip_complete_hermitian(ctx_, CP, uplo, (int)N_, dH_, (long)ldd_h_), "complete_hermitian");
            hv_valid_ = false;
            return;
        }
        const bool up = (uplo == 'U' || uplo == 'u');
        for (std::size_t j = 0; j < N_; ++j)
            for (std::size_t i = 0; i < j; ++i) {
                if (up) H_[j + i * ldh_] = conj_(H_[i + j * ldh_]);
                else    H_[i + j * ldh_] = conj_(H_[j + i * ldh_]);
            }
        h_resident_ = false;
    }

    // ---- solver life cycle -------------------------------------------------------------------------------------
    void Start() override { locked_ = 0; }

    // reference: chase_cpu.hpp:296-327 (mt19937(1337), column-major fill) + chase_gpu.hpp:520-537 (copy, H2D)
    void initVecs(bool random) override
    {
        CHASE_PHASE(ctx_, "initVecs");
        hv_valid_ = false;
        if (random && device_rng_) {
            // ChASEGPU behaviour: generate on the device (chase_gpu.hpp:520-525), no host staging of N x nevex
            hip_ok(chase_hip_fill_normal(ctx_, CP, (int)N_, (int)nevex_, dV1_, (long)N_, 0, 0, (long)N_, 1337ull), "fill_normal");
        } else {
            if (random) {
                std::mt19937 gen(1337.0);
                std::normal_distribution<> d;
                for (std::size_t j = 0; j < nevex_; ++j)
                    for (std::size_t i = 0; i < N_; ++i) V1_[i + j * ldv_] = rnd(d, gen);
            }
            hip_ok(chase_hip_upload_matrix(ctx_, CP, (int)N_, (int)nevex_, V1_, (long)ldv_, dV1_, (long)N_), "upload V");
        }
        hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, (int)nevex_, dV1_, (long)N_, dV2_, (long)N_), "lacpy");
        reset_perm();
        upload_H();
    }

    // reference: chase_gpu.hpp:1010-1018 / chase_cpu.hpp:834-841
    // chase_cpu.hpp:329-349: re-randomise the given columns (offset by fixednev) of V1 from mt19937(4242) and mirror to V2
    void ReinitColumns(std::size_t fixednev, std::size_t const* col_indices, std::size_t n_indices) override
    {
        CHASE_PHASE(ctx_, "ReinitColumns");
        if (n_indices == 0) return;
        flush_swaps();
        hv_valid_ = false;
        std::mt19937 gen(4242);
        std::normal_distribution<> d;
        std::vector<T> h(N_);
        for (std::size_t c = 0; c < n_indices; ++c) {
            const std::size_t j = fixednev + col_indices[c];
            if (j >= nevex_) throw std::invalid_argument("ReinitColumns: column out of range");
            for (auto& x : h) x = rnd(d, gen);
            hip_ok(chase_hip_upload_matrix(ctx_, CP, (int)N_, 1, h.data(), (long)N_, dV1_ + j * N_, (long)N_), "upload column");
            hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, 1, dV1_ + j * N_, (long)N_, dV2_ + j * N_, (long)N_), "lacpy");
        }
    }

    void End() override
    {
        CHASE_PHASE(ctx_, "End");
        flush_swaps();
        hip_ok(chase_hip_download_matrix(ctx_, CP, (int)N_, (int)nevex_, dV1_, (long)N_, V1_, (long)ldv_), "download V");
    }

    // ---- filter --------------------------------------------------------------------------------------------------
    void FilterPhaseStart() override
    {
        roctx_push("chase:Filter");
        flush_swaps();                                        // keep the permutation launches out of the timed phase
        chase_hip_ctx_set_phase(ctx_, 1);
        hip_ok(chase_hip_timer_start(ctx_), "timer");
    }
    void FilterPhaseEnd() override
    {
        float ms = 0;
        hip_ok(chase_hip_timer_stop(ctx_, &ms), "timer");     // synchronises the stream
        chase_hip_ctx_set_phase(ctx_, 0);
        filter_ms_ += ms;
        roctx_pop(ctx_);
    }
    // true: initVecs(random) draws N(0,1) on the device like ChASEGPU; false (default): mt19937(1337) on the host,
    // bitwise the start vectors of ChASECPU (used by the parity tests)
    void set_device_rng(bool f) override { device_rng_ = f; }
    void reset_counters() override { filter_ms_ = 0; hemm_calls_ = 0; hemm_reused_vecs_ = 0; }
    std::size_t hemm_calls() const override { return hemm_calls_; }
    std::size_t hemm_reused_vecs() const override { return hemm_reused_vecs_; }

    void Shift(T c, bool = false) override
    {
        hv_shift_ += std::real(c);          // the cached product belongs to the unshifted matrix: (H + sI) V = H V + s V
        hip_ok(chase_hip_shift_diag(ctx_, CP, (int)N_, dH_, (long)ldd_h_, std::real(c)), "shift_diag");
    }

    // V2[:, c0:c0+ncols] = alpha * H * V1[:, c0:...] + beta * V2[:, ...], c0 = locked + offset_left; then V1 <-> V2
    void HEMM(std::size_t block, T alpha, T beta, std::size_t offset_left, std::size_t offset_right = 0) override
    {
        flush_swaps();
        const std::size_t ncols = (offset_right < block) ? block - offset_right : 0;
        if (ncols != 0) {
            const std::size_t c0 = locked_ + offset_left;
            if (hv_valid_ && beta == T(0) && std::imag(alpha) == 0.0 && c0 >= hv_locked_ &&
                c0 + ncols <= hv_locked_ + hv_block_) {
                // first Chebyshev step on the Ritz vectors RR just produced: alpha (H + sI) V1 = alpha (HV + s V1) with the
                // H V that RR left behind (W A) - three streaming passes over N x ncols instead of an N^2 x ncols HEMM
                T* v1 = dV1_ + c0 * N_;
                T* v2 = dV2_ + c0 * N_;
                hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, (int)ncols, dHV_ + c0 * N_, (long)N_, v2, (long)N_), "lacpy");
                if (hv_shift_ != 0.0) {
                    const double sc[2] = {hv_shift_, 0.0};
                    hip_ok(chase_hip_memcpy_h2d(ctx_, dScal_, sc, sizeof sc), "h2d");
                    hip_ok(chase_hip_col_axpy(ctx_, CP, (int)N_, (int)ncols, (const double*)dScal_, 0, 0, 1.0, v1, (long)N_, v2,
                                              (long)N_), "axpy shift");
                }
                hip_ok(chase_hip_scale_rows(ctx_, CP, (int)N_, (int)ncols, v2, (long)N_, 0, std::real(alpha)), "scale");
                hemm_reused_vecs_ += ncols;
            } else {
                gemm('N', N_, ncols, N_, alpha, dH_, ldd_h_, dV1_ + c0 * N_, N_, beta, dV2_ + c0 * N_, N_);
                ++hemm_calls_;
            }
        }
        hv_valid_ = false;
        std::swap(dV1_, dV2_);
    }

    // ---- QR (chase_cpu.hpp:590-776) --------------------------------------------------------------------------------
    void QR(std::size_t, R cond) override
    {
        CHASE_PHASE(ctx_, "QR");
        flush_swaps(); hv_valid_ = false;
        hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, (int)locked_, dV1_, (long)N_, dV2_, (long)N_), "lacpy");
        int disable = config_.DoCholQR() ? 0 : 1;
        if (const char* s = std::getenv("CHASE_DISABLE_CHOLQR")) disable = std::atoi(s);
        R thld_hi = 1e8, thld_lo = 2e1;
        if (const char* s = std::getenv("CHASE_CHOLQR1_THLD")) thld_lo = std::atof(s);
        last_qr_variant_ = 0;
        if (disable == 1 && cond != (R)1.0) {
            householder();
        } else {
            const int variant = (cond > thld_hi) ? 3 : (cond < thld_lo ? 1 : 2);
            last_qr_variant_ = variant;
            const int info = chase_hip_cholqr(ctx_, CP, (int)N_, (int)nevex_, dV1_, (long)N_, dA_, (long)nevex_,
                                              variant, (long)N_);
            hip_ok(info, "cholqr");
            if (info != 0) householder();
        }
        hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, (int)locked_, dV2_, (long)N_, dV1_, (long)N_), "lacpy");
    }
    int last_qr_variant() const override { return last_qr_variant_; }

    // ---- Rayleigh-Ritz (cpu/rayleighRitz.hpp:61-112 + chase_cpu.hpp:778-798) -----------------------------------------
    void RR(R* ritzv, std::size_t block) override
    {
        CHASE_PHASE(ctx_, "RR");
        flush_swaps();
        T* Q = dV1_ + locked_ * N_;
        T* W = dV2_ + locked_ * N_;
        chase_hip_ctx_set_phase(ctx_, 2);                                           // H-times-block product outside the filter
        gemm('C', N_, block, N_, T(1), dH_, ldd_h_, Q, N_, T(0), W, N_);          // W = H^H Q
        chase_hip_ctx_set_phase(ctx_, 0);
        // A = W^H Q = Q^H H Q is Hermitian: tiles on and above the diagonal only (the reference's gemm computes all of it and
        // heevd reads one triangle, cpu/rayleighRitz.hpp:96-104)
        hip_ok(chase_hip_herkx(ctx_, CP, (int)block, (int)N_, W, (long)N_, Q, (long)N_, dA_, (long)block, 1), "herkx");
        hip_ok(chase_hip_heevd(ctx_, CP, (int)block, dA_, (long)block, ritzv), "heevd");
        hv_valid_ = false;
        if (resd_reuse_) {
            // H (Q A) = (H Q) A: the product the residual step needs costs N*block^2 here instead of N^2*block there
            // (the reference recomputes H V in Resd, chase_cpu.hpp:805-818; equal up to rounding)
            if (!dHV_) alloc((void**)&dHV_, N_ * nevex_ * sizeof(T));
            gemm('N', N_, block, block, T(1), W, N_, dA_, block, T(0), dHV_ + locked_ * N_, N_);
            hv_valid_ = true; hv_locked_ = locked_; hv_block_ = block; hv_shift_ = 0.0;
        }
        gemm('N', N_, block, block, T(1), Q, N_, dA_, block, T(0), W, N_);         // W = Q A
        std::swap(dV1_, dV2_);
    }

    // ---- residuals (cpu/residuals.hpp:56-81 + chase_cpu.hpp:805-818) ------------------------------------------------
    void Resd(R* ritzv, R* resd, std::size_t) override
    {
        CHASE_PHASE(ctx_, "Resd");
        flush_swaps();
        const std::size_t sub = nevex_ - locked_;
        T* V = dV1_ + locked_ * N_;
        T* W = dV2_ + locked_ * N_;
        const bool reused = hv_valid_ && hv_shift_ == 0.0 && hv_locked_ == locked_ && hv_block_ == sub;
        if (reused) W = dHV_ + locked_ * N_;                                   // H V left behind by RR
        else {
            chase_hip_ctx_set_phase(ctx_, 2);
            gemm('N', N_, sub, N_, T(1), dH_, ldd_h_, V, N_, T(0), W, N_);
            chase_hip_ctx_set_phase(ctx_, 0);
        }
        hip_ok(chase_hip_resid_norms(ctx_, CP, (int)N_, (int)sub, W, (long)N_, V, (long)N_, ritzv, resd, 0), "resid");
        if (reused || CP) recheck_borderline(ritzv, resd, sub, reused);
        if (resd != resid_.data() + locked_) std::memcpy(resid_.data() + locked_, resd, sub * sizeof(R));
    }

    // Lock on what the reference would see.  The residuals above come from products that differ from the reference's fresh
    // H v (chase_cpu.hpp:805-818) by rounding: (H Q) A instead of H (Q A), three real products per complex one.  That is
    // ~1e-14 ||H|| absolute - invisible except to a pair whose residual sits ON the tolerance, where it decides whether
    // locking() takes the pair now or one iteration later.  Every residual within 1e-3 of tol is therefore taken again from
    // a fresh four-product H v of that column (a handful of columns: one pass over H), and that value is what the driver
    // sees.  CHASE_HIP_RESD_RECHECK=0 turns it off.
    void recheck_borderline(const R* ritzv, R* resd, std::size_t sub, bool reused)
    {
        static const bool on = [] { const char* e = std::getenv("CHASE_HIP_RESD_RECHECK"); return e ? std::atoi(e) != 0 : true; }();
        if (!on) return;
        const R tol = (R)config_.GetTol();
        // the window: 1e-3 tol, or - for a tight tolerance / a large norm - the scale of the rounding gap itself, which does not
        // depend on tol (measured at config 4: 4e-15 = 0.2 eps ||H||; ||H|| from the Lanczos upper bound; the advisor's finding)
        // capped at tol / 2 (round-5 advisor): for ||H|| >= 1e5 tol / eps the rounding-gap term alone would exceed tol and every
        // converged residual would be re-taken in every iteration
        const R window = std::min(std::max((R)1e-3 * tol, (R)4 * std::numeric_limits<R>::epsilon() * norm_h_), (R)0.5 * tol);
        std::vector<std::size_t> idx;
        if (forced_recheck_ >= 0) {                           // single-rank replay: as many as the recording re-took
            for (std::size_t j = 0; j < std::min<std::size_t>((std::size_t)forced_recheck_, sub); ++j) idx.push_back(j);
        } else {
            for (std::size_t j = 0; j < sub; ++j)
                if (std::abs(resd[j] - tol) <= window) idx.push_back(j);
        }
        if (idx.empty()) return;
        const std::size_t k = idx.size();
        (void)reused;
        T* Vs = dV2_ + locked_ * N_;
        if (2 * k > sub) {
            // no room for copies beside the products: the reference's residual step as it stands on ALL unlocked columns
            // (chase_cpu.hpp:805-818, fresh four-product H V) - V2's columns are free scratch here in either case
            chase_hip_ctx_set_phase(ctx_, 3);
            gemm('N', N_, sub, N_, T(1), dH_, ldd_h_, dV1_ + locked_ * N_, N_, T(0), Vs, N_);
            chase_hip_ctx_set_phase(ctx_, 0);
            hip_ok(chase_hip_resid_norms(ctx_, CP, (int)N_, (int)sub, Vs, (long)N_, dV1_ + locked_ * N_, (long)N_, ritzv, resd, 0), "resid");
            resd_rechecked_ += sub;
            return;
        }
        // scratch: the columns of V2 behind the locked ones (free whenever RR's product is reused; otherwise they hold the
        // product this call just made, which nobody reads again)
        T* Ws = Vs + k * N_;
        std::vector<R> lam(k), out(k);
        for (std::size_t i = 0; i < k; ++i) {
            lam[i] = ritzv[idx[i]];
            hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, 1, dV1_ + (locked_ + idx[i]) * N_, (long)N_, Vs + i * N_, (long)N_), "lacpy");
        }
        chase_hip_ctx_set_phase(ctx_, 3);
        gemm('N', N_, k, N_, T(1), dH_, ldd_h_, Vs, N_, T(0), Ws, N_);
        chase_hip_ctx_set_phase(ctx_, 0);
        hip_ok(chase_hip_resid_norms(ctx_, CP, (int)N_, (int)k, Ws, (long)N_, Vs, (long)N_, lam.data(), out.data(), 0), "resid");
        for (std::size_t i = 0; i < k; ++i) resd[idx[i]] = out[i];
        resd_rechecked_ += k;
    }
    void set_forced_recheck(long k) override { forced_recheck_ = k; }
    std::size_t resd_rechecked() const override { return resd_rechecked_; }

    void recompute_residuals(std::size_t ncols, const double* lambda, double* out) override
    {
        CHASE_PHASE(ctx_, "recompute_residuals");
        if (ncols > nevex_) throw std::invalid_argument("recompute_residuals: more columns than the Impl holds");
        flush_swaps();
        hv_valid_ = false;                                                   // dV2_ is scratch from here on
        chase_hip_ctx_set_phase(ctx_, 3);                                    // four real products per complex product, always
        gemm('N', N_, ncols, N_, T(1), dH_, ldd_h_, dV1_, N_, T(0), dV2_, N_);
        chase_hip_ctx_set_phase(ctx_, 0);
        hip_ok(chase_hip_resid_norms(ctx_, CP, (int)N_, (int)ncols, dV2_, (long)N_, dV1_, (long)N_, lambda, out, 0), "resid");
    }

    void Swap(std::size_t i, std::size_t j) override
    {
        if (i == j) return;
        std::swap(perm_[i], perm_[j]);
        perm_dirty_ = true;
    }
    void Lock(std::size_t new_converged) override { locked_ += new_converged; }

    // ---- Lanczos (cpu/lanczos.hpp:46-209, :216-300) -------------------------------------------------------------------
    void Lanczos(std::size_t m, R* upperb) override
    {
        CHASE_PHASE(ctx_, "Lanczos");
        lanczosIter_ = m; numLanczos_ = 1;
        std::vector<R> theta(m);
        lanczos_core(m, 1, false, upperb, theta.data(), nullptr, nullptr);
    }
    void Lanczos(std::size_t M, std::size_t numvec, R* upperb, R* ritzv, R* Tau, R* ritzV) override
    {
        CHASE_PHASE(ctx_, "Lanczos");
        lanczosIter_ = M; numLanczos_ = numvec;
        lanczos_core(M, numvec, true, upperb, ritzv, Tau, ritzV);
    }
    // V1[:, :idx] <- V1[:, :m] * ritzVc[:, :idx]   (chase_cpu.hpp:368-382, incl. its m-column copy-back)
    void LanczosDos(std::size_t idx, std::size_t m, T* ritzVc) override
    {
        CHASE_PHASE(ctx_, "LanczosDos");
        hv_valid_ = false;
        flush_swaps();
        hip_ok(chase_hip_upload_matrix(ctx_, CP, (int)m, (int)idx, ritzVc, (long)m, dA_, (long)m), "upload ritzV");
        gemm('N', N_, idx, m, T(1), dV1_, N_, dA_, m, T(0), dV2_, N_);
        hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, (int)m, dV2_, (long)N_, dV1_, (long)N_), "lacpy");
    }

private:
    static T rnd(std::normal_distribution<>& d, std::mt19937& g)
    {
        if constexpr (is_cplx<T>::value) { const double re = d(g); const double im = d(g); return T(re, im); }
        else return T(d(g));
    }
    static T conj_(T x) { if constexpr (is_cplx<T>::value) return std::conj(x); else return x; }

    void alloc(void** p, std::size_t bytes)
    {
        int rc = chase_hip_malloc(ctx_, p, bytes);
        if (rc) throw HipStatusError(rc, "chase_hip_malloc");
        owned_.push_back(*p);
    }
    void upload_H()
    {
        if (!h_on_device_)
            hip_ok(chase_hip_upload_matrix(ctx_, CP, (int)N_, (int)N_, H_, (long)ldh_, dH_, (long)ldd_h_), "upload H");
        h_resident_ = true;
    }
    void gemm(char op, std::size_t m, std::size_t n, std::size_t k, T alpha, const T* A, std::size_t lda, const T* B,
              std::size_t ldb, T beta, T* C, std::size_t ldc)
    {
        int rc;
        if constexpr (is_cplx<T>::value) {
            const double a[2] = {alpha.real(), alpha.imag()}, b[2] = {beta.real(), beta.imag()};
            rc = chase_hip_gemm_z(ctx_, op, (int)m, (int)n, (int)k, a, A, (long)lda, B, (long)ldb, b, C, (long)ldc);
        } else {
            rc = chase_hip_gemm_d(ctx_, op, (int)m, (int)n, (int)k, alpha, A, (long)lda, B, (long)ldb, beta, C, (long)ldc);
        }
        hip_ok(rc, "gemm");
    }
    void householder()
    {
        last_qr_variant_ = 0;
        hip_ok(chase_hip_houseqr(ctx_, CP, (int)N_, (int)nevex_, dV1_, (long)N_), "houseqr");
    }
    void reset_perm()
    {
        for (std::size_t i = 0; i < nevex_; ++i) perm_[i] = (int)i;
        perm_dirty_ = false;
    }
    // apply the deferred swaps: V1'[:, j] = V1[:, perm[j]]; V2 is free scratch whenever swaps are pending (see header)
    void flush_swaps()
    {
        if (!perm_dirty_) return;
        std::vector<int> src, dst;
        for (std::size_t j = 0; j < nevex_; ++j)
            if (perm_[j] != (int)j) { src.push_back(perm_[j]); dst.push_back((int)j); }
        if (!src.empty()) {
            hip_ok(chase_hip_permute_cols(ctx_, CP, (int)N_, dV1_, (long)N_, dV2_, (long)N_, src.data(), dst.data(),
                                          (int)src.size()), "permute_cols");
            if (hv_valid_)       // the cached H V follows its vectors
                hip_ok(chase_hip_permute_cols(ctx_, CP, (int)N_, dHV_, (long)N_, dV2_, (long)N_, src.data(), dst.data(),
                                              (int)src.size()), "permute_cols");
        }
        reset_perm();
    }

    void lanczos_core(std::size_t M, std::size_t nv, bool store, R* upperb, R* theta, R* Tau, R* ritzV)
    {
        flush_swaps(); hv_valid_ = false;
        if (!h_resident_) upload_H();
        constexpr int E = CP ? 2 : 1;
        T *v0, *v1, *v2;
        double *d_alpha, *d_beta, *d_nrm0;
        void* blk = nullptr;
        const std::size_t vec_bytes = 3 * N_ * nv * sizeof(T);
        const std::size_t sc_bytes = (M * nv * E + M * nv + nv) * sizeof(double);
        int rc = chase_hip_malloc(ctx_, &blk, vec_bytes + sc_bytes);
        if (rc) throw HipStatusError(rc, "lanczos workspace");
        v0 = (T*)blk; v1 = v0 + N_ * nv; v2 = v1 + N_ * nv;
        d_alpha = (double*)(v2 + N_ * nv); d_beta = d_alpha + M * nv * E; d_nrm0 = d_beta + M * nv;
        const int n = (int)N_, nvi = (int)nv;
        try {
            hip_ok(chase_hip_memset(ctx_, blk, 0, vec_bytes + sc_bytes), "memset");
            hip_ok(chase_hip_lacpy(ctx_, CP, n, nvi, dV1_, (long)N_, v1, (long)N_), "lacpy");
            hip_ok(chase_hip_col_nrm2(ctx_, CP, n, nvi, v1, (long)N_, d_nrm0), "nrm2");
            hip_ok(chase_hip_col_scal(ctx_, CP, n, nvi, d_nrm0, 1, v1, (long)N_), "scal");
            for (std::size_t k = 0; k < M; ++k) {
                // the reference writes every run's k-th vector into column k (last run wins), cpu/lanczos.hpp:85-88
                if (store)
                    hip_ok(chase_hip_lacpy(ctx_, CP, n, 1, v1 + (nv - 1) * N_, (long)N_, dV1_ + k * N_, (long)N_), "lacpy");
                gemm('C', N_, nv, N_, T(1), dH_, ldd_h_, v1, N_, T(0), v2, N_);
                double* ak = d_alpha + k * nv * E;
                hip_ok(chase_hip_col_dot(ctx_, CP, n, nvi, v1, (long)N_, v2, (long)N_, ak), "dot");
                hip_ok(chase_hip_col_axpy(ctx_, CP, n, nvi, ak, 0, 1, -1.0, v1, (long)N_, v2, (long)N_), "axpy");
                if (k > 0)
                    hip_ok(chase_hip_col_axpy(ctx_, CP, n, nvi, d_beta + (k - 1) * nv, 1, 1, -1.0, v0, (long)N_, v2,
                                              (long)N_), "axpy");
                hip_ok(chase_hip_col_nrm2(ctx_, CP, n, nvi, v2, (long)N_, d_beta + k * nv), "nrm2");
                if (k == M - 1) break;
                hip_ok(chase_hip_col_scal(ctx_, CP, n, nvi, d_beta + k * nv, 1, v2, (long)N_), "scal");
                T* t = v0; v0 = v1; v1 = v2; v2 = t;       // (v0, v1, v2) <- (v1, v2, v0)
            }
            if (store) hip_ok(chase_hip_lacpy(ctx_, CP, n, nvi, v1, (long)N_, dV1_, (long)N_), "lacpy");
            std::vector<double> h_alpha(M * nv * E), h_beta(M * nv);
            hip_ok(chase_hip_memcpy_d2h(ctx_, h_alpha.data(), d_alpha, h_alpha.size() * sizeof(double)), "d2h");
            hip_ok(chase_hip_memcpy_d2h(ctx_, h_beta.data(), d_beta, h_beta.size() * sizeof(double)), "d2h");
            chase_hip_free(ctx_, blk);
            blk = nullptr;

            std::vector<double> d(M), e(M), w(M), Z(M * M);
            R ub = 0;
            for (std::size_t i = 0; i < nv; ++i) {
                for (std::size_t k = 0; k < M; ++k) {
                    d[k] = h_alpha[(k * nv + i) * E];                    // real(alpha)
                    e[k] = (k + 1 < M) ? h_beta[k * nv + i] : 0.0;
                }
                hip_ok(chase_hip_stemr_host((int)M, d.data(), e.data(), w.data(), Z.data(), (int)M), "stemr");
                for (std::size_t k = 0; k < M; ++k) {
                    theta[k + i * M] = w[k];
                    if (Tau) Tau[k + i * M] = std::abs(Z[k * M]) * std::abs(Z[k * M]);
                }
                if (ritzV) std::memcpy(ritzV, Z.data(), M * M * sizeof(double));   // last run wins, like the reference
                const R cand = std::max(std::abs(w[0]), std::abs(w[M - 1])) + std::abs(h_beta[(M - 1) * nv + i]);
                ub = (i == 0) ? cand : std::max(ub, cand);
            }
            *upperb = ub;
            norm_h_ = std::abs(ub);
        } catch (...) {
            if (blk) chase_hip_free(ctx_, blk);
            throw;
        }
    }

    chase_hip_ctx* ctx_;
    std::size_t N_, nev_, nex_, nevex_;
    T* H_; std::size_t ldh_;
    T* V1_; std::size_t ldv_;
    R* ritzv_;
    bool h_on_device_, own_h_ = false, h_resident_ = false;
    ConfigT config_;
    std::vector<R> resid_, early_;
    std::vector<int> perm_;
    bool perm_dirty_ = false;
    std::size_t locked_ = 0, lanczosIter_ = 0, numLanczos_ = 0;
    T *dH_ = nullptr, *dV1_ = nullptr, *dV2_ = nullptr, *dA_ = nullptr, *dHV_ = nullptr;
    std::size_t ldd_h_ = 0;
    bool hv_valid_ = false, resd_reuse_ = std::getenv("CHASE_HIP_RESD_REUSE") ? std::atoi(std::getenv("CHASE_HIP_RESD_REUSE")) != 0 : true;
    std::size_t hv_locked_ = 0, hv_block_ = 0, hemm_reused_vecs_ = 0;
    double hv_shift_ = 0.0;
    void* dScal_ = nullptr;
    std::vector<void*> owned_;
    double filter_ms_ = 0;
    std::size_t hemm_calls_ = 0, resd_rechecked_ = 0;
    long forced_recheck_ = -1;            // set_forced_recheck (single-rank replay)
    R norm_h_ = 0;                        // Lanczos upper bound of the last solve (recheck window)
    bool device_rng_ = false;
    int last_qr_variant_ = 0;
};

} // namespace chase_amd
