// pchase_hip_pseudo_impl.hpp — pChaseHipPseudo<T, BaseT>: multi-GPU Impl of the ChaseBase<T> surface for
// pseudo-Hermitian (Bethe-Salpeter) matrices on the 2D process grid (BASELINE config 5).
//
// Mirrors pChASECPU / pChASEGPU instantiated with PseudoHermitianBlockBlockMatrix / PseudoHermitianBlockCyclicMatrix:
//   constructor, initVecs      Impl/pchase_cpu/pchase_cpu.hpp:92-190,273-311     (2*(nev+nex) columns, lower rows x 0.001)
//   HEMM product               linalg/internal/mpi/hemm.hpp:112-199              (H V = S H^H S V around the conj-trans GEMM)
//   HEMM_H2                    Impl/pchase_cpu/pchase_cpu.hpp:497-548            (V2 = alpha H (H V1) + beta V2 + gamma V1)
//   ApplyKconjugate            Impl/pchase_cpu/pchase_cpu.hpp:550-570, distMultiVector.hpp:1879 (Kconjugate)
//   QR (S-orthogonal locking)  Impl/pchase_cpu/pchase_cpu.hpp:572-867
//   RR -> rayleighRitz_v2      linalg/internal/mpi/pseudo_hermitian_rayleighRitz.hpp:270-451
//   Resd                       Impl/pchase_cpu/pchase_cpu.hpp:903-915 (mpi/residuals.hpp with the pseudo product)
//   Lanczos (S inner product)  linalg/internal/mpi/pseudo_hermitian_lanczos.hpp:57-470
// Everything that is layout plumbing (redistribution, packed Gram all-reduce, agreement collectives, CholQR, the
// distributed Householder fallback, deferred Swap) is inherited from pChaseHip; the sign flips act on the rows whose
// GLOBAL index is >= N/2, so block and block-cyclic layouts are both covered.
#pragma once
#include "pchase_hip_impl.hpp"

namespace chase_amd {

template <class T, class BaseT = ChaseBase<T>, class ConfigT = ChaseConfig<T>>
class pChaseHipPseudo : public pChaseHip<T, BaseT, ConfigT> {
    using P = pChaseHip<T, BaseT, ConfigT>;
public:
    using R = Base<T>;
    using P::CP;
    using P::E;

    // H_loc: DEVICE pointer to this rank's block of the pseudo-Hermitian H; ritzv: 2*(nev+nex) reals
    pChaseHipPseudo(chase_hip_ctx* ctx, chase_hip_grid* grid, std::size_t N, std::size_t nev, std::size_t nex,
                    std::size_t mb, std::size_t nb, T* H_loc, std::size_t ldh, R* ritzv)
        : P(ctx, grid, N, nev, nex, mb, nb, H_loc, ldh, ritzv, 2 * (nev + nex))
    {
        if (N % 2) throw std::invalid_argument("pChaseHipPseudo: N must be even (2 x 2 block structure)");
        this->pseudo_ = true;
        this->alloc((void**)&dG_, this->m_ * this->n_ * sizeof(T));       // S H_loc S, rebuilt from H_loc at every initVecs
        this->dHbac_ = dG_; this->ldhbac_ = this->m_;
        build_g();
        this->alloc((void**)&dScal_, 4096);
        build_kconj_exchange();
    }

    bool isSym() override { return false; }
    bool isPseudoHerm() override { return true; }
    bool checkSymmetryEasy() override { return false; }
    bool checkPseudoHermicityEasy() override { return true; }
    void Shift(T, bool = false) override {}                   // the H^2 filter carries the shift in gamma
    void symOrHermMatrix(char) override {}          // a pseudo-Hermitian matrix has no triangle to complete (chase_cpu.hpp pseudo: no-op)
    void HEMM(std::size_t, T, T, std::size_t, std::size_t = 0) override
    {
        throw std::logic_error("pChaseHipPseudo: the pseudo-Hermitian filter uses HEMM_H2");
    }

    // V2[cols] = alpha H (H V1[cols]) + beta V2[cols] + gamma V1[cols]; the roles of V1 and V2 alternate (the reference
    // toggles next_, here the two buffers trade names so that "V1" is always the current iterate)
    void HEMM_H2(std::size_t block, T alpha, T beta, T gamma, std::size_t offset_left, std::size_t offset_right = 0) override
    {
        this->flush_swaps();
        std::size_t ncols = (offset_right < block) ? block - offset_right : 0;
        if (ncols != 0) {
            const std::size_t c0 = offset_left + this->locked_, m = this->m_;
            // the reference's filter call runs `block` columns from c0, i.e. past the first half into second-half columns
            // that ApplyKconjugate overwrites right after the filter (algorithm.inc:1012-1064): stop at the first half
            if (c0 >= this->nevex_) { std::swap(this->dV1_, this->dV2_); return; }
            if (c0 + ncols > this->nevex_) ncols = this->nevex_ - c0;
            this->hemm_ptr(true, this->dV1_, this->dW1_, c0, ncols, T(1), T(0), true);       // W1 = H V1   (row-type)
            T* v1 = this->dV1_ + c0 * m;
            T* v2 = this->dV2_ + c0 * m;
            if (std::imag(beta) == 0.0 && std::imag(gamma) == 0.0) {
                // beta V2 + gamma V1 is formed on the root of the row group BEFORE the product (which then accumulates with
                // beta' = 1 there, 0 elsewhere): nothing has to wait for the all-reduce of V2, the panel pipeline keeps
                // running across filter steps
                if (this->mycol_ == 0) {
                    const double br = std::real(beta), gr = std::real(gamma);
                    if (br == 0.0) {
                        lacpy(ncols, v1, v2);
                        hip_ok(chase_hip_scale_rows(this->ctx_, CP, (int)m, (int)ncols, v2, (long)m, 0, gr), "scale");
                    } else {
                        hip_ok(chase_hip_scale_rows(this->ctx_, CP, (int)m, (int)ncols, v2, (long)m, 0, br), "scale");
                        upload_scalar(gamma);
                        hip_ok(chase_hip_col_axpy(this->ctx_, CP, (int)m, (int)ncols, (const double*)dScal_, 0, 0, 1.0, v1, (long)m,
                                                  v2, (long)m), "axpy gamma");
                    }
                }
                this->hemm_ptr(false, this->dW1_, this->dV2_, c0, ncols, alpha, T(1), true);  // V2 += alpha H W1
            } else {
                this->hemm_ptr(false, this->dW1_, this->dV2_, c0, ncols, alpha, beta, true);  // V2 = alpha H W1 + beta V2
                this->sync_comm();
                upload_scalar(gamma);
                hip_ok(chase_hip_col_axpy(this->ctx_, CP, (int)m, (int)ncols, (const double*)dScal_, 0, 0, 1.0, v1, (long)m, v2,
                                          (long)m), "axpy gamma");
            }
            this->hemm_calls_ += 2;
        }
        std::swap(this->dV1_, this->dV2_);
    }

    // second half [nc - locked - block, nc - locked) = K-conjugate of the first half [locked, locked + block):
    // second[g, j] = conj(first[(g + N/2) mod N, j]).  The partner rows live on other grid rows: pairwise exchange inside
    // the column group like the reference's MPI_Sendrecv / ncclSendrecvWrapper pairs (distMultiVector.hpp:1879-2060,
    // grid/nccl_utils.hpp:271) - step s sends the rows whose partner lives s grid rows further to that rank and receives
    // from the rank s rows back; N even makes the row pairing an involution, so the two lists of a pair match.
    void ApplyKconjugate(std::size_t block) override
    {
        CHASE_PHASE(this->ctx_, "ApplyKconjugate");
        this->flush_swaps(); this->sync_comm();
        if (block == 0) return;
        if (block > this->nevex_) throw std::invalid_argument("ApplyKconjugate: block larger than nev+nex");
        const std::size_t m = this->m_;
        const std::size_t c2 = this->nc_ - this->locked_ - block;
        T* first = this->dV1_ + this->locked_ * m;
        T* second = this->dV1_ + c2 * m;
        const int p = this->nprow_, me = this->myrow_;
        for (int s = 0; s < p; ++s) {
            const int to = (me + s) % p, from = (me - s + p) % p;
            const KX& snd = kx_[to];
            const KX& rcv = kx_[from];
            if (snd.send_cnt)
                hip_ok(chase_hip_rows_indexed(this->ctx_, CP, first, (long)m, this->dStage_, snd.send_cnt, snd.d_send, snd.send_cnt,
                                              (int)block, 0), "kconj pack");
            T* rbuf = (s == 0) ? this->dStage_ : dRecv_;
            if (s != 0)
                P::coll(chase_hip_grid_sendrecv(this->grid_, CHASE_HIP_COL, this->dStage_, (std::size_t)snd.send_cnt * block * E,
                                                snd.send_cnt ? to : -1, dRecv_, (std::size_t)rcv.recv_cnt * block * E,
                                                rcv.recv_cnt ? from : -1));
            if (rcv.recv_cnt)
                hip_ok(chase_hip_rows_indexed(this->ctx_, CP, rbuf, rcv.recv_cnt, second, (long)m, rcv.d_recv, rcv.recv_cnt,
                                              (int)block, 1), "kconj unpack");
        }
        if (CP) hip_ok(chase_hip_conj(this->ctx_, (int)m, (int)block, second, (long)m), "conj");
        hip_ok(chase_hip_lacpy(this->ctx_, CP, (int)m, (int)block, second, (long)m, this->dV2_ + c2 * m, (long)m), "lacpy");
    }

    void QR(std::size_t, R cond) override
    {
        CHASE_PHASE(this->ctx_, "QR");
        this->flush_swaps(); this->sync_comm();
        const std::size_t L = this->locked_, m = this->m_, nc = this->nc_;
        lacpy(L, this->dV1_, this->dV2_);                                        // V2[:, :L] = V1[:, :L]
        lacpy(L, this->dV1_ + (nc - L) * m, this->dV2_ + L * m);                 // V2[:, L:2L] = V1[:, nc-L:]
        lacpy(nc - 2 * L, this->dV1_ + L * m, this->dV2_ + 2 * L * m);           // V2[:, 2L:] = active block
        std::swap(this->dV1_, this->dV2_);
        this->flip_coltype(this->dV1_, 2 * L);                                   // S-orthogonalise against the locked vectors
        int disable = this->config_.DoCholQR() ? 0 : 1;
        if (const char* s = std::getenv("CHASE_DISABLE_CHOLQR")) disable = std::atoi(s);
        R thld_hi = 1e8, thld_lo = 2e1;
        if (const char* s = std::getenv("CHASE_CHOLQR1_THLD")) thld_lo = std::atof(s);
        this->last_qr_variant_ = 0;
        if (this->forced_qr_ >= 0) {                            // single-rank replay: the recording's variant (set_forced_qr)
            if (this->forced_qr_ == 0) this->householder();
            else { this->last_qr_variant_ = this->forced_qr_; this->cholqr_dist(this->forced_qr_); }
        } else if (disable == 1 && cond != (R)1.0) {
            this->householder();
        } else {
            const int variant = (cond > thld_hi) ? 3 : (cond < thld_lo ? 1 : 2);
            this->last_qr_variant_ = variant;
            if (this->cholqr_dist(variant) != 0) this->householder();
        }
        lacpy(nc - 2 * L, this->dV1_ + 2 * L * m, this->dV2_ + L * m);           // active block back to the middle
        std::swap(this->dV1_, this->dV2_);
        lacpy(L, this->dV1_, this->dV2_);
        lacpy(L, this->dV1_ + (nc - L) * m, this->dV2_ + (nc - L) * m);
        lacpy(nc - 2 * L, this->dV1_ + L * m, this->dV2_ + L * m);
    }

    void RR(R* ritzv, std::size_t block) override
    {
        CHASE_PHASE(this->ctx_, "RR");
        this->flush_swaps(); this->sync_comm();
        const std::size_t n = 2 * block, c0 = this->locked_, m = this->m_, nl = this->n_;
        T* V1 = this->dV1_ + c0 * m;
        T* V2 = this->dV2_ + c0 * m;
        T* W1 = this->dW1_ + c0 * nl;
        T* W2 = this->dW2_ + c0 * nl;
        T* A = this->dA_;
        T* M = this->dA_ + n * n;
        P::coll(chase_hip_grid_bcast(this->grid_, CHASE_HIP_ROW, V1, m * n * E, 0, 0));
        lacpy(n, V1, V2);
        chase_hip_ctx_set_phase(this->ctx_, 2);
        this->hemm_ptr(true, this->dV1_, this->dW1_, c0, n, T(1), T(0), false);          // W1 = H Q      (row-type)
        chase_hip_ctx_set_phase(this->ctx_, 0);
        this->redistribute_c2r(V2, W2, n);                                                // W2 = Q        (row-type)
        this->flip_rowtype(W1, n);                                                        // W1 = S H Q
        this->gemm('C', n, n, nl, T(1), W2, nl, W1, nl, T(0), A, n);                      // A = Q^H S H Q
        this->allreduce_packed_upper(A, n, CHASE_HIP_ROW);
        this->flip_coltype(V1, n);                                                        // V1 = S Q
        this->gemm('C', n, n, m, T(1), V2, m, V1, m, T(0), M, n);                         // M = Q^H S Q
        this->allreduce_packed_upper(M, n, CHASE_HIP_COL);
        int info = chase_hip_pseudo_rr_small(this->ctx_, CP, (int)n, A, M, ritzv);
        P::coll(chase_hip_grid_agree_max(this->grid_, &info));
        if (info > 0 && this->replay_tolerant_) {
            // replay only (set_replay_tolerant): the lone rank's A is a partial sum and need not be positive definite - the same
            // dense core once more on A = I (kept out of the operator log: the real rank has no such second pass)
            struct Mute { chase_hip_ctx* c; Mute(chase_hip_ctx* x) : c(x) { chase_hip_ctx_oplog_mute(c, 1); } ~Mute() { chase_hip_ctx_oplog_mute(c, -1); } } mute(this->ctx_);
            hip_ok(chase_hip_set_identity(this->ctx_, CP, (int)n, A, (long)n), "set_identity");
            // (M = Q^H S Q is untouched: the dense core returns before it writes M when the factorisation of A fails)
            info = chase_hip_pseudo_rr_small(this->ctx_, CP, (int)n, A, M, ritzv);
            ++this->replay_tolerated_;
        }
        if (info > 0) throw std::runtime_error("pChaseHipPseudo::RR: Q^H S H Q is not positive definite (potrf info " +
                                               std::to_string(info) + ")");
        hip_ok(info, "pseudo_rr_small");
        this->agree_vector(ritzv, n, M, n * n);
        this->gemm('N', m, n / 2, n, T(1), V2, m, M, n, T(0), V1, m);                     // first n/2 Ritz vectors
        lacpy(n, V1, V2);                                                                 // pchase_cpu.hpp:881-883
    }

    void Lanczos(std::size_t m, R* upperb) override
    {
        CHASE_PHASE(this->ctx_, "Lanczos");
        this->lanczosIter_ = m; this->numLanczos_ = 1;
        std::vector<R> theta(m), tau(m), z(m * m);
        pseudo_lanczos(m, 1, false, theta.data(), tau.data(), z.data());
        if (upperb) *upperb = theta[m - 1];                   // mpi/pseudo_hermitian_lanczos.hpp:470
    }
    void Lanczos(std::size_t M, std::size_t numvec, R* upperb, R* ritzv, R* Tau, R* ritzV) override
    {
        CHASE_PHASE(this->ctx_, "Lanczos");
        this->lanczosIter_ = M; this->numLanczos_ = numvec;
        pseudo_lanczos(M, numvec, true, ritzv, Tau, ritzV);
        if (upperb) *upperb = ritzv[M - 1];                   // mpi/pseudo_hermitian_lanczos.hpp:295
    }

protected:
    void init_vecs_hook(bool random) override
    {
        build_g();                                                         // H_loc is caller-owned: pick up any change
        if (random) this->flip_coltype(this->dV1_, this->nc_, 0.001);      // pchase_cpu.hpp:283-300
    }

private:
    // G_loc = S H_loc S: copy H_loc, flip the local rows with global index >= N/2, then the local columns with global
    // index >= N/2 (entries in the lower-right quadrant are flipped twice, i.e. kept)
    void build_g()
    {
        const std::size_t m = this->m_, n = this->n_, half = this->N_ / 2;
        hip_ok(chase_hip_lacpy(this->ctx_, CP, (int)m, (int)n, this->dH_, (long)this->ldh_, dG_, (long)m), "lacpy");
        hip_ok(chase_hip_scale_rows_bc(this->ctx_, CP, (int)m, (int)n, dG_, (long)m, (long)half, this->Rr_.nb, this->nprow_,
                                       this->myrow_, -1.0), "flip rows");
        const long nb = this->Cc_.nb;
        for (std::size_t l0 = 0; l0 < n; l0 += (std::size_t)nb) {               // one column block at a time
            const std::size_t w = std::min<std::size_t>((std::size_t)nb, n - l0);
            const long g0 = this->Cc_.global((long)l0, this->mycol_);
            if ((std::size_t)g0 + w <= half) continue;
            const std::size_t skip = (std::size_t)g0 >= half ? 0 : half - (std::size_t)g0;   // block straddling N/2
            hip_ok(chase_hip_scale_rows(this->ctx_, CP, (int)m, (int)(w - skip), dG_ + (l0 + skip) * m, (long)m, 0, -1.0),
                   "flip columns");
        }
    }
    void lacpy(std::size_t ncols, const T* src, T* dst)
    {
        if (ncols) hip_ok(chase_hip_lacpy(this->ctx_, CP, (int)this->m_, (int)ncols, src, (long)this->m_, dst, (long)this->m_), "lacpy");
    }
    void upload_scalar(T v)
    {
        double h[2] = {std::real(v), CP ? std::imag(v) : 0.0};
        hip_ok(chase_hip_memcpy_h2d(this->ctx_, dScal_, h, sizeof h), "h2d");
    }

    // mpi/pseudo_hermitian_lanczos.hpp:57-470.  Per-step scalars go through the host (M <= 50 steps, numvec <= 16); the
    // products, redistributions and updates stay on the device; the S-inner products are summed over the column group.
    void pseudo_lanczos(std::size_t M, std::size_t nv, bool store, R* theta, R* Tau, R* ritzV)
    {
        this->flush_swaps(); this->sync_comm();
        using C = std::complex<double>;
        const std::size_t m = this->m_, nl = this->n_;
        chase_hip_ctx* ctx = this->ctx_;
        T *v0, *v1, *v2, *Sv, *vw, *tmp;
        double* dsc;
        void* blk = nullptr;
        const std::size_t vb = (5 * m + nl) * nv * sizeof(T), sb = 8 * nv * sizeof(double);
        int rc = chase_hip_malloc(ctx, &blk, vb + sb);
        if (rc) throw HipStatusError(rc, "lanczos workspace");
        v0 = (T*)blk; v1 = v0 + m * nv; v2 = v1 + m * nv; Sv = v2 + m * nv; tmp = Sv + m * nv; vw = tmp + m * nv;
        dsc = (double*)(vw + nl * nv);
        const int ml = (int)m, nvi = (int)nv;
        std::vector<double> hd(nv * 2), hc(nv * 2);
        auto dots = [&](const T* x, const T* y, std::vector<C>& out) {          // out[i] = Re(x_i^H y_i), summed over the column group
            hip_ok(chase_hip_col_dot(ctx, CP, ml, nvi, x, (long)m, y, (long)m, dsc), "dot");
            this->colgroup_sum(dsc, nv * E);
            hip_ok(chase_hip_memcpy_d2h(ctx, hd.data(), dsc, nv * E * sizeof(double)), "d2h");
            for (std::size_t i = 0; i < nv; ++i) out[i] = C(CP ? hd[2 * i] : hd[i], 0);
        };
        auto axpy = [&](const std::vector<C>& a, const T* x, T* y) {             // y_i += a_i x_i
            for (std::size_t i = 0; i < nv; ++i) { if (CP) { hc[2 * i] = a[i].real(); hc[2 * i + 1] = a[i].imag(); } else hc[i] = a[i].real(); }
            hip_ok(chase_hip_memcpy_h2d(ctx, dsc + 2 * nv, hc.data(), nv * E * sizeof(double)), "h2d");
            hip_ok(chase_hip_col_axpy(ctx, CP, ml, nvi, dsc + 2 * nv, 0, 1, 1.0, x, (long)m, y, (long)m), "axpy");
        };
        auto scal = [&](const std::vector<C>& a, T* x) {                         // x_i *= a_i  ==  x += (a - 1) x
            std::vector<C> am(nv);
            for (std::size_t i = 0; i < nv; ++i) am[i] = a[i] - C(1, 0);
            hip_ok(chase_hip_lacpy(ctx, CP, ml, nvi, x, (long)m, tmp, (long)m), "lacpy");
            axpy(am, tmp, x);
        };
        auto hv = [&]() {                                                        // v2 = H v1 ; Sv = S v2
            this->hemm_ptr(true, v1, vw, 0, nv, T(1), T(0), false);
            this->redistribute_r2c(vw, v2, nv);
            hip_ok(chase_hip_lacpy(ctx, CP, ml, nvi, v2, (long)m, Sv, (long)m), "lacpy");
            this->flip_coltype(Sv, nv);
        };
        try {
            hip_ok(chase_hip_memset(ctx, blk, 0, vb + sb), "memset");
            hip_ok(chase_hip_lacpy(ctx, CP, ml, nvi, this->dV1_, (long)m, v1, (long)m), "lacpy");
            std::vector<C> alpha(nv), beta(nv);
            std::vector<double> d(M * nv, 0.0), e(M * nv, 0.0);
            hv();
            dots(v1, Sv, beta);
            for (auto& b : beta) b = C(1, 0) / std::sqrt(b);
            scal(beta, v1); scal(beta, v2);
            for (std::size_t k = 0; k < M; ++k) {
                if (store)
                    hip_ok(chase_hip_lacpy(ctx, CP, ml, 1, v1 + (nv - 1) * m, (long)m, this->dV1_ + k * m, (long)m), "lacpy");
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
            if (store) hip_ok(chase_hip_lacpy(ctx, CP, ml, nvi, v1, (long)m, this->dV1_, (long)m), "lacpy");
            hip_ok(chase_hip_ctx_sync(ctx), "sync");
            chase_hip_free(ctx, blk);
            blk = nullptr;
            std::vector<double> dd(M), ee(M), w(M), Z(M * M);
            bool broke = false;
            for (std::size_t i = 0; i < nv; ++i) {
                for (std::size_t k = 0; k < M; ++k) { dd[k] = d[k + M * i]; ee[k] = (k + 1 < M) ? e[k + M * i] : 0.0; }
                const int rc_stemr = chase_hip_stemr_host((int)M, dd.data(), ee.data(), w.data(), Z.data(), (int)M);
                if (rc_stemr != 0 && this->replay_tolerant_) {              // replay: the tape's numbers are what the driver reads
                    std::fill(w.begin(), w.end(), 0.0); std::fill(Z.begin(), Z.end(), 0.0); ++this->replay_tolerated_;
                    broke = true;
                } else hip_ok(rc_stemr, "stemr");
                for (std::size_t k = 0; k < M; ++k) {
                    theta[k + i * M] = w[k];
                    if (Tau) Tau[k + i * M] = std::abs(Z[k * M]) * std::abs(Z[k * M]);
                }
                if (ritzV) std::memcpy(ritzV, Z.data(), M * M * sizeof(double));
            }
            if (broke && store) {
                // replay only: a recurrence normalised by partial sums can overflow; the Lanczos vectors it left in the block
                // are replaced by N(0,1) so that what follows runs on finite numbers (not in the operator log)
                struct Mute { chase_hip_ctx* c; Mute(chase_hip_ctx* x) : c(x) { chase_hip_ctx_oplog_mute(c, 1); } ~Mute() { chase_hip_ctx_oplog_mute(c, -1); } } mute(ctx);
                hip_ok(chase_hip_fill_normal(ctx, CP, ml, (int)std::min<std::size_t>(std::max(M, nv), this->nc_), this->dV1_, (long)m, 0, 0,
                                             (long)m, 4242ULL), "fill_normal");
            }
            std::vector<R> pack(theta, theta + M * nv);                         // identical spectral estimates on every rank
            if (Tau) pack.insert(pack.end(), Tau, Tau + M * nv);
            if (ritzV) pack.insert(pack.end(), ritzV, ritzV + M * M);
            this->agree_vector(pack.data(), pack.size(), nullptr, 0);
            std::size_t o = 0;
            std::memcpy(theta, pack.data() + o, M * nv * sizeof(R)); o += M * nv;
            if (Tau) { std::memcpy(Tau, pack.data() + o, M * nv * sizeof(R)); o += M * nv; }
            if (ritzV) std::memcpy(ritzV, pack.data() + o, M * M * sizeof(R));
        } catch (...) {
            if (blk) chase_hip_free(ctx, blk);
            throw;
        }
    }

    T* dG_ = nullptr;
    // K-conjugate exchange lists per member q of my column group: my local rows whose partner row lives on q (sent in
    // ascending local order), and for q's rows whose partner is mine (in q's ascending local order) my local row of that partner
    struct KX { int send_cnt = 0, recv_cnt = 0; int* d_send = nullptr; int* d_recv = nullptr; };
    std::vector<KX> kx_;
    T* dRecv_ = nullptr;
    void build_kconj_exchange()
    {
        const long N = (long)this->N_, half = N / 2;
        const int p = this->nprow_, me = this->myrow_;
        kx_.assign((std::size_t)p, KX());
        std::size_t max_recv = 1;
        for (int q = 0; q < p; ++q) {
            std::vector<int> snd, rcv;
            for (long l = 0; l < this->Rr_.count(me); ++l) {
                const long partner = (this->Rr_.global(l, me) + half) % N;
                if (this->Rr_.owner(partner) == q) snd.push_back((int)l);
            }
            for (long l = 0; l < this->Rr_.count(q); ++l) {
                const long partner = (this->Rr_.global(l, q) + half) % N;
                if (this->Rr_.owner(partner) == me) rcv.push_back((int)this->Rr_.local(partner));
            }
            kx_[(std::size_t)q].send_cnt = (int)snd.size();
            kx_[(std::size_t)q].recv_cnt = (int)rcv.size();
            if (!snd.empty()) kx_[(std::size_t)q].d_send = this->upload_ints(snd);
            if (!rcv.empty()) kx_[(std::size_t)q].d_recv = this->upload_ints(rcv);
            max_recv = std::max(max_recv, rcv.size());
        }
        this->alloc((void**)&dRecv_, max_recv * this->nevex_ * sizeof(T));
    }
    void* dScal_ = nullptr;
};

} // namespace chase_amd
