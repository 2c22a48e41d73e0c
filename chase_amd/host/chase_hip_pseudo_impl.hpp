// chase_hip_pseudo_impl.hpp — ChaseHipPseudo<T, BaseT>: single-MI355X Impl of the ChaseBase<T> surface for
// pseudo-Hermitian (Bethe-Salpeter) matrices  H = [[A, B], [-conj(B), -conj(A)]],  S H Hermitian, S = diag(I, -I).
//
// Mirrors ChASECPU<T, PseudoHermitianMatrix<T>> / ChASEGPU<T, PseudoHermitianMatrix<T, GPU>> virtual by virtual:
//   constructor, initVecs       Impl/chase_cpu/chase_cpu.hpp:78-97,296-327        (2*(nev+nex) columns, lower rows x 0.001)
//   HEMM_H2                     Impl/chase_cpu/chase_cpu.hpp:510-555               (Vec2 = alpha H (H Vec1) + beta Vec2 + gamma Vec1)
//   ApplyKconjugate             Impl/chase_cpu/chase_cpu.hpp:557-588               (second half = K-conjugate of the first)
//   QR (S-orthogonal locking)   Impl/chase_cpu/chase_cpu.hpp:590-776
//   RR -> rayleighRitz_v2       linalg/internal/cpu/rayleighRitz.hpp:285-392
//   Resd                        Impl/chase_cpu/chase_cpu.hpp:805-818
//   Lanczos (S inner product)   linalg/internal/cpu/lanczos.hpp:302-470
// Column layout of the vector block: [locked | first half (unconverged) | second half (unconverged) | locked].
// All O(N n) / O(N^2 n) work runs in the gfx950 kernels through the C ABI; the (2 nevex)^2 dense core of the
// Rayleigh-Ritz step (potrf, three trsm, heevd) runs on the host like the reference CPU path.
#pragma once
#include <cmath>
#include <complex>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/chase_hip.h"
#include "chase_hip_impl.hpp"
#include "interface.hpp"
#include "output_override.hpp"
#include "roctx.hpp"

namespace chase_amd {

template <class T, class BaseT = ChaseBase<T>, class ConfigT = ChaseConfig<T>>
class ChaseHipPseudo : public WithOutput<BaseT>, public HipImplExtras {
public:
    using R = Base<T>;
    static constexpr int CP = is_cplx<T>::value ? 1 : 0;
    static constexpr int E = CP ? 2 : 1;

    // H: N x N (ldh), V1: N x 2*(nev+nex) (ldv), ritzv: 2*(nev+nex) reals — the reference's pseudo constructor contract
    ChaseHipPseudo(chase_hip_ctx* ctx, std::size_t N, std::size_t nev, std::size_t nex, T* H, std::size_t ldh, T* V1,
                   std::size_t ldv, R* ritzv, bool h_on_device = false)
        : ctx_(ctx), N_(N), nev_(nev), nex_(nex), nevex_(nev + nex), nc_(2 * (nev + nex)), H_(H), ldh_(ldh), V1_(V1),
          ldv_(ldv), ritzv_(ritzv), h_on_device_(h_on_device), config_(N, nev, nex), resid_(2 * (nev + nex), 0),
          perm_(2 * (nev + nex))
    {
        if (!ctx) throw std::invalid_argument("ChaseHipPseudo: null context");
        if (N == 0 || nevex_ == 0 || nc_ > N) throw std::invalid_argument("ChaseHipPseudo: need 0 < 2(nev+nex) <= N");
        if (N % 2) throw std::invalid_argument("ChaseHipPseudo: N must be even (2 x 2 block structure)");
        if (ldh < N || ldv < N) throw std::invalid_argument("ChaseHipPseudo: leading dimension smaller than N");
        reset_perm();
        if (h_on_device_) { dH_ = H; ldd_h_ = ldh; }
        else { alloc((void**)&dH_, N_ * N_ * sizeof(T)); ldd_h_ = N_; }
        alloc((void**)&dV1_, N_ * nc_ * sizeof(T));
        alloc((void**)&dV2_, N_ * nc_ * sizeof(T));
        alloc((void**)&dTmp_, N_ * nc_ * sizeof(T));
        alloc((void**)&dA_, 3 * nc_ * nc_ * sizeof(T));
        alloc((void**)&dScal_, 4096);
    }
    ~ChaseHipPseudo() override { for (void* p : owned_) chase_hip_free(ctx_, p); }

    std::size_t GetN() const override { return N_; }
    std::size_t GetNev() override { return nev_; }
    std::size_t GetNex() override { return nex_; }
    std::size_t GetLanczosIter() override { return lanczosIter_; }
    std::size_t GetNumLanczos() override { return numLanczos_; }
    std::size_t GetRitzvBlockSize() const override { return nc_; }
    R* GetRitzv() override { return ritzv_; }
    R* GetResid() override { return resid_.data(); }
    ConfigT& GetConfig() override { return config_; }
    int get_nprocs() override { return 1; }
    int get_rank() override { return 0; }
    bool isSym() override { return false; }
    bool isPseudoHerm() override { return true; }
    bool checkSymmetryEasy() override { return false; }
    bool checkPseudoHermicityEasy() override { return true; }
    void symOrHermMatrix(char) override {}
    void Sort(R*, R*, R*) override {}
    void Shift(T, bool = false) override {}                   // the H^2 filter carries the shift in gamma
    void HEMM(std::size_t, T, T, std::size_t, std::size_t = 0) override
    {
        throw std::logic_error("ChaseHipPseudo: the pseudo-Hermitian filter uses HEMM_H2");
    }
    void set_early_locked_residuals(std::vector<R> r) override { early_ = std::move(r); }

    std::size_t locked() const override { return locked_; }
    int last_qr_variant() const override { return last_qr_variant_; }
    double filter_ms() const override { return filter_ms_; }
    std::size_t hemm_calls() const override { return hemm_calls_; }
    void set_device_rng(bool f) override { device_rng_ = f; }
    void reset_counters() override { filter_ms_ = 0; hemm_calls_ = 0; }
    void* device_V1() override { flush_swaps(); return dV1_; }
    std::size_t local_rows() const override { return N_; }

    void Start() override { locked_ = 0; }

    void initVecs(bool random) override
    {
        CHASE_PHASE(ctx_, "initVecs");
        if (random && device_rng_) {
            hip_ok(chase_hip_fill_normal(ctx_, CP, (int)N_, (int)nc_, dV1_, (long)N_, 0, 0, (long)N_, 1337ull), "fill_normal");
        } else {
            if (random) {
                std::mt19937 gen(1337.0);
                std::normal_distribution<> d;
                for (std::size_t j = 0; j < nc_; ++j)
                    for (std::size_t i = 0; i < N_; ++i) V1_[i + j * ldv_] = rnd(d, gen);
            }
            hip_ok(chase_hip_upload_matrix(ctx_, CP, (int)N_, (int)nc_, V1_, (long)ldv_, dV1_, (long)N_), "upload V");
        }
        if (random)   // chase_cpu.hpp:310-321: damp the lower block of the start vectors
            hip_ok(chase_hip_scale_rows(ctx_, CP, (int)N_, (int)nc_, dV1_, (long)N_, (int)(N_ / 2), 0.001), "scale_rows");
        hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, (int)nc_, dV1_, (long)N_, dV2_, (long)N_), "lacpy");
        reset_perm();
        if (!h_on_device_)
            hip_ok(chase_hip_upload_matrix(ctx_, CP, (int)N_, (int)N_, H_, (long)ldh_, dH_, (long)ldd_h_), "upload H");
    }
    // chase_cpu.hpp:329-349: re-randomise the given columns (offset by fixednev) of V1 from mt19937(4242) and mirror to V2
    void ReinitColumns(std::size_t fixednev, std::size_t const* col_indices, std::size_t n_indices) override
    {
        CHASE_PHASE(ctx_, "ReinitColumns");
        if (n_indices == 0) return;
        flush_swaps();
        
        std::mt19937 gen(4242);
        std::normal_distribution<> d;
        std::vector<T> h(N_);
        for (std::size_t c = 0; c < n_indices; ++c) {
            const std::size_t j = fixednev + col_indices[c];
            if (j >= nc_) throw std::invalid_argument("ReinitColumns: column out of range");
            for (auto& x : h) x = rnd(d, gen);
            hip_ok(chase_hip_upload_matrix(ctx_, CP, (int)N_, 1, h.data(), (long)N_, dV1_ + j * N_, (long)N_), "upload column");
            hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, 1, dV1_ + j * N_, (long)N_, dV2_ + j * N_, (long)N_), "lacpy");
        }
    }

    void End() override
    {
        CHASE_PHASE(ctx_, "End");
        flush_swaps();
        hip_ok(chase_hip_download_matrix(ctx_, CP, (int)N_, (int)nc_, dV1_, (long)N_, V1_, (long)ldv_), "download V");
    }

    void FilterPhaseStart() override
    {
        roctx_push("chase:Filter");
        flush_swaps();
        chase_hip_ctx_set_phase(ctx_, 1);
        hip_ok(chase_hip_timer_start(ctx_), "timer");
    }
    void FilterPhaseEnd() override
    {
        float ms = 0;
        hip_ok(chase_hip_timer_stop(ctx_, &ms), "timer");
        chase_hip_ctx_set_phase(ctx_, 0);
        filter_ms_ += ms;
        roctx_pop(ctx_);
    }

    void HEMM_H2(std::size_t block, T alpha, T beta, T gamma, std::size_t offset_left, std::size_t offset_right = 0) override
    {
        flush_swaps();
        std::size_t ncols = (offset_right < block) ? block - offset_right : 0;
        if (ncols != 0) {
            const std::size_t c0 = offset_left + locked_;
            // the reference's filter call runs `block` columns from c0, i.e. past the first half into second-half columns
            // that ApplyKconjugate overwrites right after the filter (algorithm.inc:1012-1064): stop at the first half
            if (c0 >= nevex_) { std::swap(dV1_, dV2_); return; }
            if (c0 + ncols > nevex_) ncols = nevex_ - c0;
            T* v1 = dV1_ + c0 * N_;
            T* v2 = dV2_ + c0 * N_;
            gemm('N', N_, ncols, N_, T(1), dH_, ldd_h_, v1, N_, T(0), dTmp_, N_);
            gemm('N', N_, ncols, N_, alpha, dH_, ldd_h_, dTmp_, N_, beta, v2, N_);
            upload_scalar(gamma);
            hip_ok(chase_hip_col_axpy(ctx_, CP, (int)N_, (int)ncols, (const double*)dScal_, 0, 0, 1.0, v1, (long)N_, v2,
                                      (long)N_), "axpy gamma");
            hemm_calls_ += 2;
        }
        std::swap(dV1_, dV2_);
    }

    void ApplyKconjugate(std::size_t block) override
    {
        CHASE_PHASE(ctx_, "ApplyKconjugate");
        flush_swaps();
        const std::size_t h = N_ / 2, c2 = nc_ - locked_ - block;
        T* first = dV1_ + locked_ * N_;
        T* second = dV1_ + c2 * N_;
        hip_ok(chase_hip_lacpy(ctx_, CP, (int)h, (int)block, first, (long)N_, second + h, (long)N_), "lacpy");
        hip_ok(chase_hip_lacpy(ctx_, CP, (int)h, (int)block, first + h, (long)N_, second, (long)N_), "lacpy");
        if (CP) hip_ok(chase_hip_conj(ctx_, (int)N_, (int)block, second, (long)N_), "conj");
    }

    void QR(std::size_t, R cond) override
    {
        CHASE_PHASE(ctx_, "QR");
        flush_swaps();
        const std::size_t L = locked_;
        const int n = (int)N_;
        lacpy(L, dV1_, dV2_);                                            // V2[:, :L] = V1[:, :L]
        lacpy(L, dV1_ + (nc_ - L) * N_, dV2_ + L * N_);                  // V2[:, L:2L] = V1[:, nc-L:]
        lacpy(nc_ - 2 * L, dV1_ + L * N_, dV2_ + 2 * L * N_);            // V2[:, 2L:] = active block
        std::swap(dV1_, dV2_);
        // S-orthogonalise against the locked vectors: flip the lower half of the 2L locked columns
        hip_ok(chase_hip_scale_rows(ctx_, CP, n, (int)(2 * L), dV1_, (long)N_, (int)(N_ / 2), -1.0), "flip");
        int disable = config_.DoCholQR() ? 0 : 1;
        if (const char* s = std::getenv("CHASE_DISABLE_CHOLQR")) disable = std::atoi(s);
        R thld_hi = 1e8, thld_lo = 2e1;
        if (const char* s = std::getenv("CHASE_CHOLQR1_THLD")) thld_lo = std::atof(s);
        last_qr_variant_ = 0;
        if (disable == 1 && cond != (R)1.0) {
            hip_ok(chase_hip_houseqr(ctx_, CP, n, (int)nc_, dV1_, (long)N_), "houseqr");
        } else {
            const int variant = (cond > thld_hi) ? 3 : (cond < thld_lo ? 1 : 2);
            last_qr_variant_ = variant;
            const int info = chase_hip_cholqr(ctx_, CP, n, (int)nc_, dV1_, (long)N_, dA_, (long)nc_, variant, (long)N_);
            hip_ok(info, "cholqr");
            if (info != 0) {
                last_qr_variant_ = 0;
                hip_ok(chase_hip_houseqr(ctx_, CP, n, (int)nc_, dV1_, (long)N_), "houseqr");
            }
        }
        lacpy(nc_ - 2 * L, dV1_ + 2 * L * N_, dV2_ + L * N_);            // active block back to the middle of V2
        std::swap(dV1_, dV2_);
        lacpy(L, dV1_, dV2_);
        lacpy(L, dV1_ + (nc_ - L) * N_, dV2_ + (nc_ - L) * N_);
    }

    void RR(R* ritzv, std::size_t block) override
    {
        CHASE_PHASE(ctx_, "RR");
        flush_swaps();
        const std::size_t n = 2 * block, k = N_ / 2;
        T* Q = dV1_ + locked_ * N_;
        T* W = dV2_ + locked_ * N_;
        T* A = dA_;
        T* M = dA_ + n * n;
        chase_hip_ctx_set_phase(ctx_, 2);
        gemm('N', N_, n, N_, T(1), dH_, ldd_h_, Q, N_, T(0), W, N_);                      // W = H Q
        chase_hip_ctx_set_phase(ctx_, 0);
        hip_ok(chase_hip_scale_rows(ctx_, CP, (int)N_, (int)n, W, (long)N_, (int)k, -1.0), "flip");   // W = S H Q
        gemm('C', n, n, N_, T(1), Q, N_, W, N_, T(0), A, n);                              // A = Q^H S H Q
        hip_ok(chase_hip_set_identity(ctx_, CP, (int)n, M, (long)n), "identity");
        gemm('C', n, n, k, T(-2), Q + k, N_, Q + k, N_, T(1), M, n);                     // M = I - 2 Q2^H Q2
        const int info = chase_hip_pseudo_rr_small(ctx_, CP, (int)n, A, M, ritzv);
        if (info > 0) throw std::runtime_error("ChaseHipPseudo::RR: Q^H S H Q is not positive definite (potrf info " +
                                               std::to_string(info) + ")");
        hip_ok(info, "pseudo_rr_small");
        gemm('N', N_, n / 2, n, T(1), Q, N_, M, n, T(0), W, N_);                          // first n/2 Ritz vectors
        std::swap(dV1_, dV2_);
    }

    void Resd(R* ritzv, R* resd, std::size_t) override
    {
        CHASE_PHASE(ctx_, "Resd");
        flush_swaps();
        const std::size_t sub = nevex_ - locked_;
        T* V = dV1_ + locked_ * N_;
        T* W = dV2_ + locked_ * N_;
        chase_hip_ctx_set_phase(ctx_, 3);                     // the residuals the convergence test reads: four products
        gemm('N', N_, sub, N_, T(1), dH_, ldd_h_, V, N_, T(0), W, N_);
        chase_hip_ctx_set_phase(ctx_, 0);
        hip_ok(chase_hip_resid_norms(ctx_, CP, (int)N_, (int)sub, W, (long)N_, V, (long)N_, ritzv, resd, 0), "resid");
        if (resd != resid_.data() + locked_) std::memcpy(resid_.data() + locked_, resd, sub * sizeof(R));
    }

    void recompute_residuals(std::size_t ncols, const double* lambda, double* out) override
    {
        CHASE_PHASE(ctx_, "recompute_residuals");
        if (ncols > 2 * nevex_) throw std::invalid_argument("recompute_residuals: more columns than the Impl holds");
        flush_swaps();
        chase_hip_ctx_set_phase(ctx_, 3);
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
    void Lock(std::size_t k) override { locked_ += k; }

    void Lanczos(std::size_t m, R* upperb) override
    {
        CHASE_PHASE(ctx_, "Lanczos");
        lanczosIter_ = m; numLanczos_ = 1;
        std::vector<R> theta(m), tau(m), z(m * m);
        lanczos_core(m, 1, false, theta.data(), tau.data(), z.data());
        if (upperb) *upperb = theta[m - 1];                   // cpu/lanczos.hpp:629: largest Ritz value of the run
    }
    void Lanczos(std::size_t M, std::size_t numvec, R* upperb, R* ritzv, R* Tau, R* ritzV) override
    {
        CHASE_PHASE(ctx_, "Lanczos");
        lanczosIter_ = M; numLanczos_ = numvec;
        lanczos_core(M, numvec, true, ritzv, Tau, ritzV);
        if (upperb) *upperb = ritzv[M - 1];                   // cpu/lanczos.hpp:515
    }
    void LanczosDos(std::size_t idx, std::size_t m, T* ritzVc) override
    {
        CHASE_PHASE(ctx_, "LanczosDos");
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
    void alloc(void** p, std::size_t bytes)
    {
        int rc = chase_hip_malloc(ctx_, p, bytes);
        if (rc) throw HipStatusError(rc, "chase_hip_malloc");
        owned_.push_back(*p);
    }
    void lacpy(std::size_t ncols, const T* src, T* dst)
    {
        if (ncols) hip_ok(chase_hip_lacpy(ctx_, CP, (int)N_, (int)ncols, src, (long)N_, dst, (long)N_), "lacpy");
    }
    void upload_scalar(T v)
    {
        double h[2] = {std::real(v), CP ? std::imag(v) : 0.0};
        hip_ok(chase_hip_memcpy_h2d(ctx_, dScal_, h, sizeof h), "h2d");
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
    void reset_perm() { for (std::size_t i = 0; i < nc_; ++i) perm_[i] = (int)i; perm_dirty_ = false; }
    void flush_swaps()
    {
        if (!perm_dirty_) return;
        std::vector<int> src, dst;
        for (std::size_t j = 0; j < nc_; ++j)
            if (perm_[j] != (int)j) { src.push_back(perm_[j]); dst.push_back((int)j); }
        if (!src.empty())
            hip_ok(chase_hip_permute_cols(ctx_, CP, (int)N_, dV1_, (long)N_, dTmp_, (long)N_, src.data(), dst.data(),
                                          (int)src.size()), "permute_cols");
        reset_perm();
    }

    // S-inner-product Lanczos (cpu/lanczos.hpp:302-470).  The per-step scalars go through the host: M <= 50 steps of
    // numvec <= 16 vectors, the products and updates stay on the device.
    void lanczos_core(std::size_t M, std::size_t nv, bool store, R* theta, R* Tau, R* ritzV)
    {
        flush_swaps();
        using C = std::complex<double>;
        T *v0, *v1, *v2, *Sv;
        double* dsc;
        void* blk = nullptr;
        const std::size_t vb = 4 * N_ * nv * sizeof(T), sb = 8 * nv * sizeof(double);
        int rc = chase_hip_malloc(ctx_, &blk, vb + sb);
        if (rc) throw HipStatusError(rc, "lanczos workspace");
        v0 = (T*)blk; v1 = v0 + N_ * nv; v2 = v1 + N_ * nv; Sv = v2 + N_ * nv;
        dsc = (double*)(Sv + N_ * nv);
        const int n = (int)N_, nvi = (int)nv;
        std::vector<double> hd(nv * 2), hc(nv * 2);
        auto dots = [&](const T* x, const T* y, std::vector<C>& out) {          // out[i] = x_i^H y_i
            hip_ok(chase_hip_col_dot(ctx_, CP, n, nvi, x, (long)N_, y, (long)N_, dsc), "dot");
            hip_ok(chase_hip_memcpy_d2h(ctx_, hd.data(), dsc, nv * E * sizeof(double)), "d2h");
            for (std::size_t i = 0; i < nv; ++i) out[i] = CP ? C(hd[2 * i], hd[2 * i + 1]) : C(hd[i], 0);
        };
        auto axpy = [&](const std::vector<C>& a, const T* x, T* y) {             // y_i += a_i x_i
            for (std::size_t i = 0; i < nv; ++i) { if (CP) { hc[2 * i] = a[i].real(); hc[2 * i + 1] = a[i].imag(); } else hc[i] = a[i].real(); }
            hip_ok(chase_hip_memcpy_h2d(ctx_, dsc + 2 * nv, hc.data(), nv * E * sizeof(double)), "h2d");
            hip_ok(chase_hip_col_axpy(ctx_, CP, n, nvi, dsc + 2 * nv, 0, 1, 1.0, x, (long)N_, y, (long)N_), "axpy");
        };
        auto scal = [&](const std::vector<C>& a, T* x) {                         // x_i *= a_i (complex scale via axpy on zeroed copy)
            // x <- a x  ==  x += (a - 1) x
            std::vector<C> am(nv);
            for (std::size_t i = 0; i < nv; ++i) am[i] = a[i] - C(1, 0);
            hip_ok(chase_hip_lacpy(ctx_, CP, n, nvi, x, (long)N_, dTmp_, (long)N_), "lacpy");
            axpy(am, dTmp_, x);
        };
        auto hv = [&]() {                                                        // v2 = H v1 ; Sv = S v2
            gemm('N', N_, nv, N_, T(1), dH_, ldd_h_, v1, N_, T(0), v2, N_);
            hip_ok(chase_hip_lacpy(ctx_, CP, n, nvi, v2, (long)N_, Sv, (long)N_), "lacpy");
            hip_ok(chase_hip_scale_rows(ctx_, CP, n, nvi, Sv, (long)N_, (int)(N_ / 2), -1.0), "flip");
        };
        try {
            hip_ok(chase_hip_memset(ctx_, blk, 0, vb + sb), "memset");
            hip_ok(chase_hip_lacpy(ctx_, CP, n, nvi, dV1_, (long)N_, v1, (long)N_), "lacpy");
            std::vector<C> alpha(nv), beta(nv);
            std::vector<double> d(M * nv, 0.0), e(M * nv, 0.0);
            hv();
            dots(v1, Sv, beta);
            for (auto& b : beta) b = C(1, 0) / std::sqrt(b);
            scal(beta, v1); scal(beta, v2);
            for (std::size_t k = 0; k < M; ++k) {
                if (store)
                    hip_ok(chase_hip_lacpy(ctx_, CP, n, 1, v1 + (nv - 1) * N_, (long)N_, dV1_ + k * N_, (long)N_), "lacpy");
                dots(v2, Sv, alpha);
                for (std::size_t i = 0; i < nv; ++i) alpha[i] = -alpha[i] * beta[i];
                axpy(alpha, v1, v2);
                for (std::size_t i = 0; i < nv; ++i) { alpha[i] = -alpha[i]; d[k + M * i] = alpha[i].real(); }
                if (k == M - 1) break;
                for (auto& b : beta) b = -C(1, 0) / b;
                axpy(beta, v0, v2);
                for (auto& b : beta) b = -b;
                T* t = v0; v0 = v1; v1 = v2; v2 = t;                            // (v0, v1, v2) <- (v1, v2, v0)
                hv();
                dots(v1, Sv, beta);
                for (std::size_t i = 0; i < nv; ++i) { beta[i] = std::sqrt(beta[i]); e[k + M * i] = beta[i].real(); beta[i] = C(1, 0) / beta[i]; }
                scal(beta, v1); scal(beta, v2);
            }
            if (store) hip_ok(chase_hip_lacpy(ctx_, CP, n, nvi, v1, (long)N_, dV1_, (long)N_), "lacpy");
            hip_ok(chase_hip_ctx_sync(ctx_), "sync");
            chase_hip_free(ctx_, blk);
            blk = nullptr;
            std::vector<double> dd(M), ee(M), w(M), Z(M * M);
            for (std::size_t i = 0; i < nv; ++i) {
                for (std::size_t k = 0; k < M; ++k) { dd[k] = d[k + M * i]; ee[k] = (k + 1 < M) ? e[k + M * i] : 0.0; }
                hip_ok(chase_hip_stemr_host((int)M, dd.data(), ee.data(), w.data(), Z.data(), (int)M), "stemr");
                for (std::size_t k = 0; k < M; ++k) {
                    theta[k + i * M] = w[k];
                    if (Tau) Tau[k + i * M] = std::abs(Z[k * M]) * std::abs(Z[k * M]);
                }
                if (ritzV) std::memcpy(ritzV, Z.data(), M * M * sizeof(double));
            }
        } catch (...) {
            if (blk) chase_hip_free(ctx_, blk);
            throw;
        }
    }

    chase_hip_ctx* ctx_;
    std::size_t N_, nev_, nex_, nevex_, nc_;
    T* H_; std::size_t ldh_;
    T* V1_; std::size_t ldv_;
    R* ritzv_;
    bool h_on_device_;
    ConfigT config_;
    std::vector<R> resid_, early_;
    std::vector<int> perm_;
    bool perm_dirty_ = false, device_rng_ = false;
    std::size_t locked_ = 0, lanczosIter_ = 0, numLanczos_ = 0;
    T *dH_ = nullptr, *dV1_ = nullptr, *dV2_ = nullptr, *dTmp_ = nullptr, *dA_ = nullptr;
    void* dScal_ = nullptr;
    std::size_t ldd_h_ = 0;
    std::vector<void*> owned_;
    double filter_ms_ = 0;
    std::size_t hemm_calls_ = 0;
    int last_qr_variant_ = 0;
};

} // namespace chase_amd
