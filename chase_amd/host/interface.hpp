// interface.hpp — the operator surface the solver driver talks to, and the solver configuration.
//
// This is the drop-in boundary.  The virtuals below have the same names, argument meaning and order
// as chase::ChaseBase<T> in the reference (algorithm/interface.hpp:46-434), so that
//   * the driver in algorithm.hpp issues exactly the reference's call sequence, and
//   * ChaseHip<T, Base> (chase_hip_impl.hpp) can be instantiated with Base = chase::ChaseBase<T> inside a real ChASE
//     checkout or with Base = chase_amd::ChaseBase<T> here, where the reference sources are absent.
// ChaseConfig mirrors the getters/setters and defaults of algorithm/configuration.hpp:31-56,174-186,644.
#pragma once
#include <complex>
#include <cstddef>
#include <limits>
#include <type_traits>
#include <vector>

namespace chase_amd {

template <class T> struct BaseOf { using type = T; };
template <class R> struct BaseOf<std::complex<R>> { using type = R; };
template <class T> using Base = typename BaseOf<T>::type;

template <class T>
class ChaseConfig {
public:
    ChaseConfig(std::size_t N, std::size_t nev, std::size_t nex) : N_(N), nev_(nev), nex_(nex)
    {
        static_assert(std::is_same<Base<T>, double>::value, "this backend is fp64 / complex-fp64 only");
    }
    std::size_t GetN() const { return N_; }
    std::size_t GetNev() const { return nev_; }
    std::size_t GetNex() const { return nex_; }

    bool UseApprox() const { return approx_; }
    void SetApprox(bool f) { approx_ = f; }
    bool DoOptimization() const { return opt_; }
    void SetOpt(bool f) { opt_ = f; }
    std::size_t GetMaxDeg() const { return max_deg_; }
    void SetMaxDeg(std::size_t d) { max_deg_ = d + d % 2; }          // degrees are kept even
    std::size_t GetDegExtra() const { return deg_extra_; }
    void SetDegExtra(std::size_t d) { deg_extra_ = d; }
    std::size_t GetMaxIter() const { return max_iter_; }
    void SetMaxIter(std::size_t m) { max_iter_ = m; }
    std::size_t GetDeg() const { return deg_; }
    void SetDeg(std::size_t d) { deg_ = d + d % 2; }
    double GetTol() const { return tol_; }
    void SetTol(double t) { tol_ = t; }
    std::size_t GetLanczosIter() const { return lanczos_iter_; }
    void SetLanczosIter(std::size_t m) { lanczos_iter_ = m; }
    std::size_t GetNumLanczos() const { return num_lanczos_; }
    void SetNumLanczos(std::size_t m) { num_lanczos_ = m; }
    bool DoCholQR() const { return cholqr_; }
    void SetCholQR(bool f) { cholqr_ = f; }
    float GetDecayingRate() const { return decaying_rate_; }
    void SetDecayingRate(float r) { decaying_rate_ = r; }
    bool UseClusterAwareDegrees() const { return cluster_aware_; }          // pseudo-Hermitian degree heuristics
    void SetClusterAwareDegrees(bool f) { cluster_aware_ = f; }
    float GetUpperbScaleRate() const { return upperb_scale_; }
    void SetUpperbScaleRate(float r) { upperb_scale_ = r; }

private:
    std::size_t N_, nev_, nex_;
    bool opt_ = true, approx_ = false, cholqr_ = true, cluster_aware_ = true;
    std::size_t max_iter_ = 25, deg_extra_ = 2, num_lanczos_ = 4;
    std::size_t max_deg_ = 36, deg_ = 20, lanczos_iter_ = 25;      // fp64 defaults
    double tol_ = 1e-10;
    float decaying_rate_ = 1.0f, upperb_scale_ = 1.0f;
};

template <class T>
class ChaseBase {
public:
    virtual ~ChaseBase() = default;
    virtual void Shift(T c, bool isunshift = false) = 0;
    virtual void HEMM(std::size_t nev, T alpha, T beta, std::size_t offset_left, std::size_t offset_right = 0) = 0;
    virtual void HEMM_H2(std::size_t nev, T alpha, T beta, T gamma, std::size_t offset_left,
                         std::size_t offset_right = 0) = 0;
    virtual void ApplyKconjugate(std::size_t block) = 0;
    virtual void FilterPhaseStart() {}
    virtual void FilterPhaseEnd() {}
    virtual void QR(std::size_t fixednev, Base<T> cond) = 0;
    virtual void RR(Base<T>* ritzv, std::size_t block) = 0;
    virtual void Sort(Base<T>* ritzv, Base<T>* residLast, Base<T>* resid) = 0;
    virtual void Resd(Base<T>* ritzv, Base<T>* resd, std::size_t fixednev) = 0;
    virtual void Lanczos(std::size_t m, Base<T>* upperb) = 0;
    virtual void Lanczos(std::size_t M, std::size_t numvec, Base<T>* upperb, Base<T>* ritzv, Base<T>* Tau,
                         Base<T>* ritzV) = 0;
    virtual void LanczosDos(std::size_t idx, std::size_t m, T* ritzVc) = 0;
    virtual void Swap(std::size_t i, std::size_t j) = 0;
    virtual void Lock(std::size_t new_converged) = 0;
    virtual bool checkSymmetryEasy() = 0;
    virtual bool isSym() = 0;
    virtual bool checkPseudoHermicityEasy() = 0;
    virtual bool isPseudoHerm() = 0;
    virtual void symOrHermMatrix(char uplo) = 0;
    virtual void Start() = 0;
    virtual void End() = 0;
    virtual void initVecs(bool random) = 0;
    virtual void ReinitColumns(std::size_t, std::size_t const*, std::size_t) {}
    virtual std::size_t GetN() const = 0;
    virtual std::size_t GetNev() = 0;
    virtual std::size_t GetNex() = 0;
    virtual std::size_t GetLanczosIter() = 0;
    virtual std::size_t GetNumLanczos() = 0;
    virtual std::size_t GetRitzvBlockSize() const = 0;
    virtual Base<T>* GetRitzv() = 0;
    virtual Base<T>* GetResid() = 0;
    virtual ChaseConfig<T>& GetConfig() = 0;
    virtual int get_nprocs() = 0;
    virtual int get_rank() = 0;
    virtual void set_early_locked_residuals(std::vector<Base<T>>) {}
};

} // namespace chase_amd
