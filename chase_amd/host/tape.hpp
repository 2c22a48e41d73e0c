// tape.hpp — record / replay of everything a ChaseBase<T> kernel tells the solver driver.
//
// The driver (algorithm.hpp; the reference's algorithm/algorithm.inc:1376-1788) steers a solve ONLY by the host-visible
// numbers the kernel hands back: Ritz values after RR, residuals after Resd, the Lanczos outputs.  TapeKernel<T> wraps any
// kernel with the ChaseBase<T> surface:
//   record  every virtual is forwarded; after RR / Resd / Lanczos the host outputs are appended to a flat tape of doubles
//           (plus, per QR, the variant the kernel took and, per Resd, how many borderline residuals it re-took);
//   replay  every virtual is forwarded to the wrapped kernel - which does ALL of its device work at ITS shapes - and the host
//           outputs are then overwritten with the taped ones, so the unmodified driver takes exactly the recorded decisions
//           (degrees, locking, bounds) and issues exactly the recorded call sequence whatever the wrapped kernel computed.
// Use: measuring ONE rank of a multi-GPU solve on a one-GPU box (bench.py --replay-rank 4x2): the tape of a real solve of
// the workload drives pChaseHip<T> on a loopback grid (chase_hip_grid_create_loopback) with that rank's local block shapes
// (Impl/pchase_gpu/pchase_gpu.hpp:1550-1700 and linalg/internal/nccl/hemm.hpp:25-399 are what such a rank executes in the
// reference).  The replayed kernel's numbers are wrong by construction; its launches, shapes and time are the real rank's.
// Both drivers (chase::Solve, chase::Solve_pseudo).  A replayed pseudo-Hermitian kernel is told to tolerate what its partial sums
// may bring (HipImplExtras::set_replay_tolerant: a projected matrix that does not factorise).
#pragma once
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "impl_extras.hpp"
#include "interface.hpp"

namespace chase_amd {

struct ScalarTape {
    enum Tag { RR = 1, RESD = 2, LANCZOS1 = 3, LANCZOSN = 4, QR = 5 };
    std::vector<double> data;          // frames: tag, count, values...
    std::size_t pos = 0;               // replay cursor
    std::size_t qr_variant_mismatches = 0;
    void rewind() { pos = 0; qr_variant_mismatches = 0; }
    void put(Tag t, const double* v, std::size_t n)
    {
        data.push_back((double)t);
        data.push_back((double)n);
        data.insert(data.end(), v, v + n);
    }
    // next frame must be (t, n): returns its values
    const double* take(Tag t, std::size_t n)
    {
        if (pos + 2 > data.size()) throw std::runtime_error("tape: replay ran past the end of the recording");
        const int tag = (int)data[pos];
        const std::size_t cnt = (std::size_t)data[pos + 1];
        if (tag == (int)t && cnt == n && pos + 2 + cnt > data.size())
            throw std::runtime_error("tape: the recording ends inside a frame (truncated tape)");
        if (tag != (int)t || cnt != n)
            throw std::runtime_error("tape: replay diverged from the recording (frame " + std::to_string(tag) + "/" +
                                     std::to_string(cnt) + ", expected " + std::to_string((int)t) + "/" + std::to_string(n) + ")");
        const double* v = data.data() + pos + 2;
        pos += 2 + cnt;
        return v;
    }
};

template <class T>
class TapeKernel : public ChaseBase<T> {
public:
    using R = Base<T>;
    enum Mode { RECORD = 1, REPLAY = 2 };
    TapeKernel(ChaseBase<T>* inner, HipImplExtras* extras, ScalarTape* tape, Mode mode)
        : k_(inner), ex_(extras), tape_(tape), mode_(mode)
    {
        static_assert(sizeof(R) == sizeof(double), "fp64 only");
        if (!inner || !tape) throw std::invalid_argument("TapeKernel: null argument");
        if (mode == RECORD) { tape->data.clear(); tape->rewind(); }
        else { tape->rewind(); if (ex_) ex_->set_replay_tolerant(true); }
    }
    ~TapeKernel() override { if (ex_) { ex_->set_forced_recheck(-1); ex_->set_forced_qr(-1); ex_->set_replay_tolerant(false); } }

    // ---- the calls whose host outputs steer the driver ------------------------------------------------------------------
    void RR(R* ritzv, std::size_t block) override
    {
        k_->RR(ritzv, block);
        // the pseudo-Hermitian RR hands back both halves of the +- spectrum (2 * block values, rayleighRitz_v2)
        exchange(ScalarTape::RR, ritzv, k_->isPseudoHerm() ? 2 * block : block);
    }
    void Resd(R* ritzv, R* resd, std::size_t fixednev) override
    {
        const std::size_t sub = k_->GetNev() + k_->GetNex() - locked_;
        double rechecked = 0;
        if (mode_ == REPLAY) {
            // frame layout: [residuals..., number of residuals the recording kernel re-took on the tolerance]
            const double* v = tape_->take(ScalarTape::RESD, sub + 1);
            if (ex_) ex_->set_forced_recheck((long)v[sub]);          // the replayed kernel re-takes as many columns
            k_->Resd(ritzv, resd, fixednev);
            if (ex_) ex_->set_forced_recheck(-1);
            std::memcpy(resd, v, sub * sizeof(double));
            return;
        }
        const std::size_t before = ex_ ? ex_->resd_rechecked() : 0;
        k_->Resd(ritzv, resd, fixednev);
        if (ex_) rechecked = (double)(ex_->resd_rechecked() - before);
        std::vector<double> f(resd, resd + sub);
        f.push_back(rechecked);
        tape_->put(ScalarTape::RESD, f.data(), f.size());
    }
    void Lanczos(std::size_t m, R* upperb) override
    {
        k_->Lanczos(m, upperb);
        exchange(ScalarTape::LANCZOS1, upperb, 1);
    }
    void Lanczos(std::size_t M, std::size_t numvec, R* upperb, R* ritzv, R* Tau, R* ritzV) override
    {
        k_->Lanczos(M, numvec, upperb, ritzv, Tau, ritzV);
        const std::size_t nt = M * numvec, n = 1 + 2 * nt + M * M;
        if (mode_ == RECORD) {
            std::vector<double> f;
            f.reserve(n);
            f.push_back(*upperb);
            f.insert(f.end(), ritzv, ritzv + nt);
            f.insert(f.end(), Tau, Tau + nt);
            f.insert(f.end(), ritzV, ritzV + M * M);
            tape_->put(ScalarTape::LANCZOSN, f.data(), n);
        } else {
            const double* v = tape_->take(ScalarTape::LANCZOSN, n);
            *upperb = v[0];
            std::memcpy(ritzv, v + 1, nt * sizeof(double));
            std::memcpy(Tau, v + 1 + nt, nt * sizeof(double));
            std::memcpy(ritzV, v + 1 + 2 * nt, M * M * sizeof(double));
        }
    }
    void QR(std::size_t fixednev, R cond) override
    {
        if (mode_ == REPLAY) {
            // the replayed kernel takes the variant the recording took (its own numbers might fail a Cholesky factorisation
            // the real rank's did not, or the other way round); what it then reports must be that variant
            const double want = *tape_->take(ScalarTape::QR, 1);
            if (ex_ && want >= 0) ex_->set_forced_qr((int)want);
            k_->QR(fixednev, cond);
            if (ex_) {
                ex_->set_forced_qr(-1);
                if ((double)ex_->last_qr_variant() != want) ++tape_->qr_variant_mismatches;
            }
            return;
        }
        k_->QR(fixednev, cond);
        double variant = ex_ ? (double)ex_->last_qr_variant() : -1.0;
        tape_->put(ScalarTape::QR, &variant, 1);
    }
    void Lock(std::size_t n) override { locked_ += n; k_->Lock(n); }
    void Start() override { locked_ = 0; k_->Start(); }

    // ---- plain forwarding ---------------------------------------------------------------------------------------------
    void Shift(T c, bool isunshift = false) override { k_->Shift(c, isunshift); }
    void HEMM(std::size_t nev, T alpha, T beta, std::size_t ol, std::size_t orr = 0) override { k_->HEMM(nev, alpha, beta, ol, orr); }
    void HEMM_H2(std::size_t nev, T a, T b, T g, std::size_t ol, std::size_t orr = 0) override { k_->HEMM_H2(nev, a, b, g, ol, orr); }
    void ApplyKconjugate(std::size_t block) override { k_->ApplyKconjugate(block); }
    void FilterPhaseStart() override { k_->FilterPhaseStart(); }
    void FilterPhaseEnd() override { k_->FilterPhaseEnd(); }
    void Sort(R* a, R* b, R* c) override { k_->Sort(a, b, c); }
    void LanczosDos(std::size_t idx, std::size_t m, T* ritzVc) override { k_->LanczosDos(idx, m, ritzVc); }
    void Swap(std::size_t i, std::size_t j) override { k_->Swap(i, j); }
    bool checkSymmetryEasy() override { return k_->checkSymmetryEasy(); }
    bool isSym() override { return k_->isSym(); }
    bool checkPseudoHermicityEasy() override { return k_->checkPseudoHermicityEasy(); }
    bool isPseudoHerm() override { return k_->isPseudoHerm(); }
    void symOrHermMatrix(char uplo) override { k_->symOrHermMatrix(uplo); }
    void End() override { k_->End(); }
    void initVecs(bool random) override { k_->initVecs(random); }
    void ReinitColumns(std::size_t f, std::size_t const* c, std::size_t n) override { k_->ReinitColumns(f, c, n); }
    std::size_t GetN() const override { return k_->GetN(); }
    std::size_t GetNev() override { return k_->GetNev(); }
    std::size_t GetNex() override { return k_->GetNex(); }
    std::size_t GetLanczosIter() override { return k_->GetLanczosIter(); }
    std::size_t GetNumLanczos() override { return k_->GetNumLanczos(); }
    std::size_t GetRitzvBlockSize() const override { return k_->GetRitzvBlockSize(); }
    R* GetRitzv() override { return k_->GetRitzv(); }
    R* GetResid() override { return k_->GetResid(); }
    ChaseConfig<T>& GetConfig() override { return k_->GetConfig(); }
    int get_nprocs() override { return k_->get_nprocs(); }
    int get_rank() override { return k_->get_rank(); }
    void set_early_locked_residuals(std::vector<R> r) override { k_->set_early_locked_residuals(std::move(r)); }

private:
    void exchange(ScalarTape::Tag t, R* v, std::size_t n)
    {
        if (mode_ == RECORD) tape_->put(t, v, n);
        else std::memcpy(v, tape_->take(t, n), n * sizeof(double));
    }
    ChaseBase<T>* k_;
    HipImplExtras* ex_;
    ScalarTape* tape_;
    Mode mode_;
    std::size_t locked_ = 0;
};

} // namespace chase_amd
