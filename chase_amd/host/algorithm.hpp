// algorithm.hpp — host-side ChASE driver: subspace iteration with a Chebyshev filter, QR, Rayleigh-Ritz, residual
// check and locking.  Scalar logic only; every touch of matrix data goes through the Kernel's ChaseBase virtuals.
//
// The reference driver (algorithm/algorithm.inc) cannot travel to the GPU box, so this is an own restatement that issues
// the SAME sequence of virtual calls with the SAME scalar arguments:
//   solve        <- algorithm/algorithm.inc:1376-1788
//   filter       <- :942-1009      (three-term Chebyshev recurrence, columns retire from the left)
//   calc_degrees <- :136-193       (per-column degree from residual and rho, even, exchange sort)
//   locking      <- :519-578       (ascending Ritz order, early-lock of stagnating pairs)
//   lanczos      <- :1067-1214     (DoS estimate of the lower filter bound)
// Kernel is any type with the ChaseBase<T> surface (chase_amd::ChaseBase<T> here, chase::ChaseBase<T> in ChASE).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>
#include "interface.hpp"

namespace chase_amd {

struct SolveStats {
    std::size_t iterations = 0;
    std::size_t filtered_vecs = 0;      // sum over HEMM calls of active columns (performance.hpp:559-563)
    std::size_t lanczos_vecs = 0;
    std::size_t locked = 0;
    double t_all = 0, t_init = 0, t_lanczos = 0, t_filter = 0, t_qr = 0, t_rr = 0, t_resid = 0;
    double lowerb = 0, upperb = 0, lambda = 0;
    std::vector<std::size_t> iter_unconverged, iter_filtered, iter_maxdeg;
};

// optional observer of the virtual-call sequence (tests compare it against the oracle's trace) and of the outer
// iterations (bench.py times single iterations through `iter_hook`: called after Lock() of every outer iteration with
// the iteration index, the filtered-vector count of that iteration and the locked / unconverged counts after it; a
// non-zero return ends the iteration loop early - the solve then finishes like one that ran into maxIter)
struct CallTrace {
    std::vector<std::string> lines;
    bool enabled = false;
    int (*iter_hook)(void* user, std::size_t iteration, std::size_t filtered, std::size_t locked, std::size_t unconverged) = nullptr;
    void* iter_user = nullptr;
    void add(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
};
inline void CallTrace::add(const char* fmt, ...)
{
    if (!enabled) return;
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    lines.emplace_back(buf);
}

template <class T, class Kernel>
class Algorithm {
public:
    using R = Base<T>;
    using clock = std::chrono::steady_clock;
    static double since(clock::time_point t0) { return std::chrono::duration<double>(clock::now() - t0).count(); }
    // The reference sizes its Lanczos runs as min(nev+nex, N/2, lanczosIter) made even and asserts m >= 1
    // (algorithm/algorithm.inc:1073,1438-1442): a problem too small for two steps is outside its domain (undefined behaviour
    // in a release build there) - refused here with a message instead
    static void require_lanczos_steps(std::size_t m)
    {
        if (m < 2) throw std::invalid_argument("chase: problem too small - the Lanczos bounds need min(nev+nex, N/2) >= 2");
    }

    // growth factor of the Chebyshev polynomial outside [-1, 1]
    static R rho_of(R t)
    {
        const R s = std::sqrt(std::abs(t * t - 1));
        return std::max(std::abs(t - s), std::abs(t + s));
    }

    static std::size_t calc_degrees(Kernel* k, std::size_t unconverged, std::size_t nex, R upperb, R lowerb, R tol,
                                    R* ritzv, R* resid, std::size_t* degrees, std::size_t locked)
    {
        auto& cfg = k->GetConfig();
        const R c = (upperb + lowerb) / 2, e = (upperb - lowerb) / 2;
        const std::size_t active = unconverged - nex;
        for (std::size_t i = 0; i < active; ++i) {
            const R rho = rho_of((ritzv[i] - c) / e);
            const std::size_t d = (std::size_t)std::ceil(std::abs(std::log(resid[i] / tol) / std::log(rho)));
            degrees[i] = std::min(d + cfg.GetDegExtra(), cfg.GetMaxDeg());
        }
        for (std::size_t i = active; i < unconverged; ++i) degrees[i] = degrees[active - 1];
        for (std::size_t i = 0; i < unconverged; ++i) degrees[i] += degrees[i] % 2;
        // exchange sort by ascending degree; the permutation is mirrored on the vectors through Swap()
        for (std::size_t j = 0; j + 1 < unconverged; ++j)
            for (std::size_t i = j; i < unconverged; ++i)
                if (degrees[i] < degrees[j]) {
                    std::swap(degrees[i], degrees[j]);
                    std::swap(ritzv[i], ritzv[j]);
                    std::swap(resid[i], resid[j]);
                    k->Swap(i + locked, j + locked);
                }
        return degrees[unconverged - 1];
    }

    static std::size_t locking(Kernel* k, std::size_t candidates, R tol, R* ritzv, R* resid, R* residLast,
                               std::vector<R>* early, std::size_t locked)
    {
        std::vector<int> order(candidates);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return ritzv[a] < ritzv[b]; });
        const bool sym = k->isSym();
        std::size_t converged = 0;
        for (std::size_t pos = 0; pos < candidates; ++pos) {
            const std::size_t j = (std::size_t)order[pos];      // NB: not refreshed after swaps (reference behaviour)
            const bool stagnating = sym && resid[j] >= residLast[j] && resid[j] < 100.0 * tol;
            if (resid[j] <= tol || stagnating) {
                if (resid[j] > tol) early->push_back(resid[j]);
                if (j != converged) {
                    std::swap(resid[j], resid[converged]);
                    std::swap(residLast[j], residLast[converged]);
                    std::swap(ritzv[j], ritzv[converged]);
                    k->Swap(j + locked, converged + locked);
                }
                ++converged;
            }
        }
        return converged;
    }

    static std::size_t filter(Kernel* k, std::size_t n, std::size_t unprocessed, std::size_t deg,
                              const std::size_t* degrees, R lambda_1, R lower, R upper, CallTrace* tr = nullptr)
    {
        const R c = (upper + lower) / 2, e = (upper - lower) / 2;
        const R sigma_1 = e / (lambda_1 - c);
        R sigma = sigma_1;
        std::size_t offset = 0, done = 0, Av = 0, next = 0;   // next: first column still being filtered

        k->FilterPhaseStart();
        k->Shift(T(-c));
        T alpha = T(sigma_1 / e), beta = T(0.0);
        if (tr) tr->add("HEMM %zu %.17g %.17g %zu", unprocessed, (double)std::real(alpha), (double)std::real(beta), offset / n);
        k->HEMM(unprocessed, alpha, beta, offset / n);
        Av += unprocessed;
        ++done;
        while (unprocessed != 0 && degrees[next] <= done) { ++next; --unprocessed; offset += n; }

        for (std::size_t i = 2; i <= deg; ++i) {
            const R sigma_new = 1.0 / (2.0 / sigma_1 - sigma);
            alpha = T(2.0 * sigma_new / e);
            beta = T(-sigma * sigma_new);
            if (tr) tr->add("HEMM %zu %.17g %.17g %zu", unprocessed, (double)std::real(alpha), (double)std::real(beta), offset / n);
            k->HEMM(unprocessed, alpha, beta, offset / n);
            sigma = sigma_new;
            Av += unprocessed;
            ++done;
            while (unprocessed != 0 && degrees[next] <= done) { ++next; --unprocessed; offset += n; }
        }
        k->Shift(T(+c), true);
        k->FilterPhaseEnd();
        return Av;
    }

    // Lanczos + density-of-states estimate of the filter's lower bound; returns the number of extracted vectors.
    static std::size_t lanczos(Kernel* k, int N, int numvec, int m, int nevex, R* upperb, bool mode, R* ritzv)
    {
        if (!mode) {
            k->Lanczos((std::size_t)m, upperb);
            return 0;
        }
        std::vector<R> Theta((std::size_t)numvec * m, 0), Tau((std::size_t)numvec * m, 0), ritzV((std::size_t)m * m, 0);
        k->Lanczos((std::size_t)m, (std::size_t)numvec, upperb, Theta.data(), Tau.data(), ritzV.data());

        std::vector<double> sorted(Theta.begin(), Theta.end());
        std::sort(sorted.begin(), sorted.end());
        const R lambda = (R)sorted[0];
        R lowerb = 0;                                   // (reference leaves it uninitialised if the scan never trips)

        const double sigma = 0.25, threshold = 2 * sigma * sigma / 10;
        const double search = (double)nevex / (double)N;
        auto G = [&](double x) { return 0.5 * (1 + std::erf(x / std::sqrt(2 * sigma * sigma))); };
        std::size_t bound = (std::size_t)m;
        if (k->isPseudoHerm()) bound /= 2;
        const long total = (long)numvec * (long)bound;
        double prev = 0;
        for (long i = 0; i + 1 < total; ++i) {
            double curr = 0;
            for (long j = 0; j < total; ++j) {
                if (sorted[i] < Theta[j] - threshold) continue;
                else if (sorted[i] > Theta[j] + threshold) curr += Tau[j];
                else curr += Tau[j] * G(sorted[i] - Theta[j]);
            }
            curr /= numvec;
            if (curr > search) {
                if (std::abs(curr - search) < std::abs(prev - search))
                    lowerb = (R)((i + 1 < total) ? sorted[i + 1] : sorted[i]);
                else
                    lowerb = (R)sorted[i];
                break;
            }
            prev = curr;
        }

        // vectors of the LAST run whose Ritz value lies below lowerb become approximate eigenvectors
        int idx = 0;
        for (int i = 0; i < m; ++i)
            if (Theta[(std::size_t)(numvec - 1) * m + i] > lowerb) { idx = i - 1; break; }
        if (idx > 0) {
            std::vector<T> ritzVc((std::size_t)m * m);
            for (std::size_t i = 0; i < ritzVc.size(); ++i) ritzVc[i] = T(ritzV[i]);
            k->LanczosDos((std::size_t)idx, (std::size_t)m, ritzVc.data());
        }
        for (int i = 0; i < idx; ++i) ritzv[i] = Theta[(std::size_t)(numvec - 1) * m + i];
        for (int i = std::max(idx, 0); i < nevex - 1; ++i) ritzv[i] = lambda;
        ritzv[nevex - 1] = lowerb;
        for (int i = 1; i < idx; ++i) {                 // intersperse the extracted vectors through the block
            const int j = i * (nevex / idx);
            k->Swap((std::size_t)i, (std::size_t)j);
            std::swap(ritzv[i], ritzv[j]);
        }
        return (std::size_t)std::max(idx, 0);
    }

    // ================= pseudo-Hermitian (BSE) path: filter on H^2, K-conjugate subspace of 2*(nev+nex) columns =========
    // restates algorithm/algorithm.inc:18-133 (detect_eigenvalue_clusters), :196-317 (calc_degrees_pseudo_H2),
    // :730-817 (locking_pseudo_v3), :1012-1064 (filter_H2), :1217-1373 (lanczos_for_H2), :1834-2220 (solve_pseudo)
    static R crho(R t)
    {
        const std::complex<R> z(t * t - 1, 0), s = std::sqrt(z), ct(t, 0);
        return std::max(std::abs(ct - s), std::abs(ct + s));
    }

    static void detect_eigenvalue_clusters(const R* ritzv, const R* resid, R tol, std::size_t unconverged, std::size_t nex,
                                           R upperb, R lowerb, std::vector<R>& f)
    {
        const std::size_t n = unconverged - nex;
        f.assign(n, 1.0);
        if (n == 0) return;
        const R thr = std::abs(upperb - lowerb) * 1e-6;
        R mean = 0;
        for (std::size_t i = 0; i < n; ++i) mean += resid[i];
        mean /= (R)n;
        std::vector<R> w(n);
        for (std::size_t i = 0; i < n; ++i) w[i] = std::min(1.0 + std::log(1.0 + resid[i] / (mean + 1e-14)), (R)2.5);
        for (std::size_t i = 0; i < n; ++i) {
            R dens = 0;
            std::size_t nb = 0;
            for (std::size_t j = 0; j < n; ++j)
                if (i != j) {
                    const R dist = std::abs(ritzv[i] - ritzv[j]);
                    if (dist < thr) { dens += w[j] / (dist + 1e-14); ++nb; }
                }
            const R spatial = nb > 0 ? 1.0 + std::log(1.0 + dens * 0.1) : 1.0;
            R comb = spatial * w[i];
            if (nb > 2 && resid[i] > 2.0 * mean) comb *= 1.2;
            if (resid[i] > 10.0 * tol) comb *= 1.15;
            f[i] = std::min((R)3.0, std::max((R)0.5, comb));
        }
        const std::vector<R> t = f;
        for (std::size_t i = 1; i + 1 < n; ++i) f[i] = 0.25 * t[i - 1] + 0.5 * t[i] + 0.25 * t[i + 1];
        for (std::size_t i = 0; i < n; ++i) f[i] = std::min((R)3.0, std::max((R)0.5, f[i]));
    }

    static std::size_t calc_degrees_pseudo_H2(Kernel* k, std::size_t unconverged, std::size_t nex, R upperb, R lowerb, R tol,
                                              R* ritzv, R* resid, const R* residLast, std::size_t* degrees,
                                              std::size_t locked)
    {
        auto& cfg = k->GetConfig();
        const std::size_t deg_extra = cfg.GetDegExtra(), deg_max = cfg.GetMaxDeg();
        std::vector<R> cf;
        if (cfg.UseClusterAwareDegrees()) detect_eigenvalue_clusters(ritzv, resid, tol, unconverged, nex, upperb, lowerb, cf);
        const R c = (upperb + lowerb) / 2, e = (upperb - lowerb) / 2;
        if (e <= 0) {
            for (std::size_t i = 0; i < unconverged; ++i) degrees[i] = deg_max + deg_max % 2;
            return deg_max + deg_max % 2;
        }
        for (std::size_t i = 0; i < unconverged; ++i) {
            const R t = (ritzv[i] * ritzv[i] - c) / e;
            const R rho = crho(t);
            std::size_t deg;
            if (!std::isfinite(rho) || rho <= 1) deg = deg_max;
            else {
                const R lr = std::log(resid[i] / tol) / std::log(rho);
                if (!std::isfinite(lr)) deg = deg_max;
                else {
                    deg = (std::size_t)std::ceil(std::abs((double)lr));
                    if (cfg.UseClusterAwareDegrees()) {
                        deg = (std::size_t)(deg * (i < cf.size() ? cf[i] : (R)1.0));
                        if (resid[i] <= tol * 10.0) {
                            const R rel = std::abs(resid[i] - residLast[i]) / (resid[i] + 1e-14);
                            if (rel < 0.1) deg += 6;
                        }
                        if (std::abs(ritzv[i]) < std::abs(upperb - lowerb) * 0.1) deg += 2;
                    }
                    deg = std::min(deg + deg_extra, deg_max);
                }
            }
            degrees[i] = deg + deg % 2;
        }
        for (std::size_t j = 0; j + 1 < unconverged; ++j)
            for (std::size_t i = j; i < unconverged; ++i)
                if (degrees[i] < degrees[j]) {
                    std::swap(degrees[i], degrees[j]);
                    std::swap(ritzv[i], ritzv[j]);
                    std::swap(resid[i], resid[j]);
                    k->Swap(i + locked, j + locked);
                }
        return *std::max_element(degrees, degrees + unconverged);
    }

    static std::size_t filter_H2(Kernel* k, std::size_t unconverged, const std::size_t* degrees, R lambda_1, R lower, R upper)
    {
        if (lower >= upper) std::swap(lower, upper);
        const R c = (upper + lower) / 2, e = (upper - lower) / 2;
        const R sigma_1 = e / (lambda_1 - c);
        R sigma = sigma_1;
        const std::size_t deg_max = *std::max_element(degrees, degrees + unconverged);
        T alpha = T(sigma_1 / e);
        k->HEMM_H2(unconverged, alpha, T(0), T(-alpha * T(c)), 0, 0);
        std::size_t Av = 2 * unconverged, s = 0;
        for (std::size_t t = 2; t <= deg_max; ++t) {
            if (s >= unconverged) break;
            const R tau = 1.0 / (2.0 / sigma_1 - sigma);
            alpha = T(2.0 * tau / e);
            k->HEMM_H2(unconverged, alpha, T(-(sigma * tau)), T(-alpha * T(c)), s, 0);
            Av += 2 * (unconverged - s);
            sigma = tau;
            while (s < unconverged && degrees[s] <= t) ++s;
        }
        return Av;
    }

    static std::size_t locking_pseudo_v3(Kernel* k, std::size_t unconverged, std::size_t nex, R tol, const std::size_t* index,
                                         R* ritzv, R* resid, R* residLast, std::vector<R>* early, std::size_t locked,
                                         std::size_t iteration)
    {
        const std::vector<R> snapshot(resid, resid + 2 * unconverged);
        std::vector<std::size_t> unconv;
        std::size_t converged = 0;
        for (std::size_t q = 0; q + nex < unconverged; ++q) {
            const std::size_t j = index[q];
            const bool stagn = resid[j] > tol && resid[j] >= residLast[q] && resid[j] <= 1000.0 * tol && iteration >= 4;
            if (resid[j] <= tol || stagn) {
                if (stagn) early->push_back(resid[j]);
                if (j != converged) {
                    std::swap(resid[j], resid[converged]);
                    std::swap(ritzv[j], ritzv[converged]);
                    k->Swap(j + locked, converged + locked);
                }
                ++converged;
            } else unconv.push_back(j);
        }
        for (std::size_t q = unconverged - nex; q < unconverged; ++q) unconv.push_back(index[q]);
        for (std::size_t i = converged; i < unconverged; ++i) residLast[i] = snapshot[unconv[i - converged]];
        return converged;
    }

    static std::size_t lanczos_for_H2(Kernel* k, int N, int numvec, int m, int nevex, R* upperb, R* ritzv)
    {
        std::vector<R> Theta((std::size_t)numvec * m, 0), Tau((std::size_t)numvec * m, 0), ritzV((std::size_t)m * m, 0);
        k->Lanczos((std::size_t)m, (std::size_t)numvec, upperb, Theta.data(), Tau.data(), ritzV.data());
        std::vector<double> srt(Theta.begin(), Theta.end());
        std::sort(srt.begin(), srt.end());
        const double sigma = 0.25, thresh = 2 * sigma * sigma / 10;
        auto G = [&](double x) { return 0.5 * (1 + std::erf(x / std::sqrt(2 * sigma * sigma))); };
        R max_abs = 0, min_abs = std::abs(Theta[0]);
        int i_min = 0;
        const int n_dos = numvec * m;
        for (int i = 0; i < n_dos; ++i) {
            const R a = std::abs(Theta[i]);
            if (a > max_abs) max_abs = a;
            if (a < min_abs) { min_abs = a; i_min = i; }
        }
        const R mu_1 = Theta[i_min] * Theta[i_min];
        *upperb = max_abs * max_abs;
        auto& cfg = k->GetConfig();
        double search_hi = ((double)N / 2 - (double)cfg.GetNev() - (double)cfg.GetNex() - 1) / (double)N;
        search_hi = std::min(1.0, std::max(0.0, search_hi));
        R lam_nn = (R)srt[n_dos - 1];
        double prev = 0;
        for (int i = 0; i < n_dos; ++i) {
            double curr = 0;
            for (int j = 0; j < n_dos; ++j) {
                if (srt[i] < Theta[j] - thresh) continue;
                else if (srt[i] > Theta[j] + thresh) curr += Tau[j];
                else curr += Tau[j] * G(srt[i] - Theta[j]);
            }
            curr /= numvec;
            if (curr > search_hi) {
                if (std::abs(curr - search_hi) < std::abs(prev - search_hi)) lam_nn = (R)srt[i];
                else lam_nn = (R)(i > 0 ? srt[i - 1] : srt[i]);
                break;
            }
            prev = curr;
            lam_nn = (R)srt[i];
        }
        const R mu_nn = lam_nn * lam_nn;
        int idx = 0;
        for (int i = 0; i < m; ++i) {
            if (Theta[(std::size_t)(numvec - 1) * m + i] > lam_nn) { idx = i - 1; break; }
            idx = i + 1;
        }
        if (idx < 0) idx = 0;
        if (idx > 0) {
            std::vector<T> ritzVc((std::size_t)m * m);
            for (std::size_t i = 0; i < ritzVc.size(); ++i) ritzVc[i] = T(ritzV[i]);
            k->LanczosDos((std::size_t)idx, (std::size_t)m, ritzVc.data());
        }
        for (int i = 0; i < idx; ++i) { const R th = Theta[(std::size_t)(numvec - 1) * m + i]; ritzv[i] = th * th; }
        for (int i = idx; i < nevex - 1; ++i) ritzv[i] = mu_1;
        ritzv[nevex - 1] = mu_nn;
        for (int i = 1; i < idx; ++i) {
            const int j = i * (nevex / idx);
            k->Swap((std::size_t)i, (std::size_t)j);
            std::swap(ritzv[i], ritzv[j]);
        }
        return (std::size_t)idx;
    }

    static void solve_pseudo(Kernel* k, SolveStats* st = nullptr, CallTrace* tr = nullptr)
    {
        SolveStats local;
        if (!st) st = &local;
        auto t_all0 = clock::now();
        auto& cfg = k->GetConfig();
        k->Start();
        const std::size_t N = cfg.GetN(), nev = cfg.GetNev(), nex = cfg.GetNex(), nevex = nev + nex;
        std::size_t unconverged = nevex;
        R* const ritzv_all = k->GetRitzv();
        R* const resid_all = k->GetResid();
        std::size_t deg = cfg.GetDeg();
        deg += deg % 2;
        deg = std::min(deg, cfg.GetMaxDeg());
        std::vector<std::size_t> degrees_all(2 * nevex, 0);
        for (std::size_t i = 0; i < unconverged; ++i) degrees_all[i] = deg;
        const double tol = cfg.GetTol();
        std::vector<R> residLast_all(2 * nevex, std::numeric_limits<R>::max());
        for (std::size_t i = 0; i < 2 * nevex; ++i) resid_all[i] = std::numeric_limits<R>::max();
        auto t0 = clock::now();
        const bool random = !cfg.UseApprox();
        k->initVecs(random);
        if (random) k->QR(0, (R)1.0);
        st->t_init = since(t0);
        t0 = clock::now();
        std::size_t lanczos_iter = std::min(nevex, std::min(N / 2, cfg.GetLanczosIter()));
        if (lanczos_iter % 2 != 0) { cfg.SetLanczosIter(lanczos_iter - 1); lanczos_iter = cfg.GetLanczosIter(); }
        require_lanczos_steps(lanczos_iter);
        R upperb = 0;
        lanczos_for_H2(k, (int)N, (int)cfg.GetNumLanczos(), (int)lanczos_iter, (int)nevex, &upperb, ritzv_all);
        st->lanczos_vecs = lanczos_iter * cfg.GetNumLanczos();
        st->t_lanczos = since(t0);
        const R mu_1 = *std::min_element(ritzv_all, ritzv_all + nevex - 1);
        const R mu_nn = ritzv_all[nevex - 1];
        upperb = upperb > 0 ? upperb * cfg.GetUpperbScaleRate() : upperb / cfg.GetUpperbScaleRate();
        const R lambda_1 = mu_1, b_sup = upperb;
        R lower = mu_nn * cfg.GetDecayingRate();
        R new_mu = mu_nn, new_l1 = lambda_1;
        if (tr) tr->add("bounds %.10e %.10e %.10e", (double)lambda_1, (double)lower, (double)b_sup);
        std::vector<R> early;
        std::vector<std::size_t> index(2 * nevex), order(nevex);
        std::size_t locked = 0, iteration = 0;
        while (locked < nev && unconverged > 0 && iteration < cfg.GetMaxIter()) {
            R* ritzv = ritzv_all + locked;
            R* resid = resid_all + locked;
            R* residLast = residLast_all.data() + locked;
            std::size_t* degrees = degrees_all.data() + locked;
            if (iteration > 0) {
                new_mu = new_mu * new_mu;
                new_l1 = new_l1 * new_l1;
                if (new_mu < lower && new_mu > lambda_1) lower = new_mu;
            }
            if (cfg.DoOptimization() && iteration != 0)
                deg = calc_degrees_pseudo_H2(k, unconverged, nex, b_sup, lower, (R)tol, ritzv, resid, residLast, degrees, locked);
            t0 = clock::now();
            if (tr) tr->add("filter it=%zu unconverged=%zu deg=%zu", iteration, unconverged, deg);
            k->FilterPhaseStart();
            const std::size_t Av_it = filter_H2(k, unconverged, degrees, lambda_1, lower, b_sup);
            st->filtered_vecs += Av_it;
            k->FilterPhaseEnd();
            st->t_filter += since(t0);
            k->ApplyKconjugate(unconverged);
            const R cc = (b_sup + lower) / 2;
            R ee = (b_sup - lower) / 2;
            if (ee <= 0) ee = std::abs(lower - b_sup) / 2;
            const R t_1 = (lambda_1 - cc) / ee;
            const R t_k = iteration > 0 ? (ritzv[0] * ritzv[0] - cc) / ee : t_1;
            const std::size_t dmax = *std::max_element(degrees, degrees + unconverged);
            const R cond = std::pow(crho(t_k), (R)degrees[0]) * std::pow(crho(t_1), (R)(dmax - degrees[0]));
            t0 = clock::now();
            if (tr) tr->add("QR %zu %.6e", locked, (double)cond);
            k->QR(locked, cond);
            st->t_qr += since(t0);
            t0 = clock::now();
            k->RR(ritzv, unconverged);
            st->t_rr += since(t0);
            t0 = clock::now();
            k->Resd(ritzv, resid, locked);
            st->t_resid += since(t0);
            std::iota(index.begin(), index.begin() + 2 * unconverged, 0);
            std::iota(order.begin(), order.begin() + unconverged, 0);
            std::sort(order.begin(), order.begin() + unconverged, [&](std::size_t a, std::size_t b) { return ritzv[a] < ritzv[b]; });
            new_mu = ritzv[order[(std::size_t)(unconverged * 0.95) - 1]] * cfg.GetDecayingRate();
            new_l1 = ritzv[order[0]];
            const std::size_t new_conv =
                locking_pseudo_v3(k, unconverged, nex, (R)tol, index.data(), ritzv, resid, residLast, &early, locked, iteration);
            if (new_conv > 0) k->ApplyKconjugate(new_conv);
            if (tr) tr->add("Lock %zu", new_conv);
            k->Lock(new_conv);
            st->iter_unconverged.push_back(unconverged);
            locked += new_conv;
            unconverged -= new_conv;
            ++iteration;
            st->iterations = iteration; st->locked = locked;
            if (tr && tr->iter_hook && tr->iter_hook(tr->iter_user, iteration - 1, Av_it, locked, unconverged)) break;
        }
        // positive Ritz values first (ascending), then the rest
        std::size_t n_re = locked + unconverged;
        if (n_re == 0) n_re = 1;
        std::vector<std::size_t> perm(n_re);
        std::iota(perm.begin(), perm.end(), 0);
        std::sort(perm.begin(), perm.end(), [&](std::size_t i, std::size_t j) {
            const bool ip = ritzv_all[i] > 0, jp = ritzv_all[j] > 0;
            if (ip != jp) return ip;
            return ritzv_all[i] < ritzv_all[j];
        });
        std::vector<bool> seen(n_re, false);
        for (std::size_t i = 0; i < n_re; ++i) {
            if (seen[i] || perm[i] == i) continue;
            std::vector<std::size_t> cyc;
            for (std::size_t cur = i; !seen[cur]; cur = perm[cur]) { seen[cur] = true; cyc.push_back(cur); }
            const R r0 = ritzv_all[cyc[0]], s0 = resid_all[cyc[0]];
            for (std::size_t q = 0; q + 1 < cyc.size(); ++q) {
                ritzv_all[cyc[q]] = ritzv_all[cyc[q + 1]];
                resid_all[cyc[q]] = resid_all[cyc[q + 1]];
                k->Swap(cyc[q], cyc[q + 1]);
            }
            ritzv_all[cyc.back()] = r0;
            resid_all[cyc.back()] = s0;
        }
        k->set_early_locked_residuals(early);
        k->End();
        st->iterations = iteration;
        st->locked = locked;
        st->lowerb = lower; st->upperb = b_sup; st->lambda = lambda_1;
        st->t_all = since(t_all0);
    }

    static void solve(Kernel* k, SolveStats* st = nullptr, CallTrace* tr = nullptr)
    {
        SolveStats local;
        if (!st) st = &local;
        auto t_all0 = clock::now();
        auto& cfg = k->GetConfig();
        k->Start();
        const std::size_t N = cfg.GetN(), nev = cfg.GetNev(), nex = cfg.GetNex(), nevex = nev + nex;
        const double tol = cfg.GetTol();
        R* const resid_all = k->GetResid();
        R* const ritzv_all = k->GetRitzv();
        std::vector<std::size_t> degrees_all(nevex);
        std::vector<R> residLast_all(nevex, std::numeric_limits<R>::max());
        std::vector<R> early;
        for (std::size_t i = 0; i < nevex; ++i) resid_all[i] = std::numeric_limits<R>::max();

        std::size_t deg = cfg.GetDeg();
        deg += deg % 2;
        deg = std::min(deg, cfg.GetMaxDeg());
        std::fill(degrees_all.begin(), degrees_all.end(), deg);

        auto t0 = clock::now();
        const bool random = !cfg.UseApprox();
        if (tr) tr->add("initVecs %d", (int)random);
        k->initVecs(random);
        if (random) {
            if (tr) tr->add("QR 0 1");
            k->QR(0, (R)1.0);
        }
        st->t_init = since(t0);

        t0 = clock::now();
        std::size_t lanczos_iter = std::min(nevex, std::min(N / 2, cfg.GetLanczosIter()));
        if (lanczos_iter % 2 != 0) {                    // even number of Ritz values (pseudo-Hermitian symmetry)
            cfg.SetLanczosIter(lanczos_iter - 1);
            lanczos_iter = cfg.GetLanczosIter();
        }
        require_lanczos_steps(lanczos_iter);
        // the numLanczos start vectors of the random-start runs are the first numLanczos columns of the block
        // (linalg/internal/cpu/lanczos.hpp:46-209 copies them without asking how many there are)
        if (random && nevex < cfg.GetNumLanczos())
            throw std::invalid_argument("chase: nev + nex must be at least numLanczos (the Lanczos runs start from that many columns)");
        R upperb = 0;
        if (tr) tr->add("Lanczos %zu %zu", lanczos_iter, cfg.GetNumLanczos());
        lanczos(k, (int)N, (int)cfg.GetNumLanczos(), (int)lanczos_iter, (int)nevex, &upperb, random,
                random ? ritzv_all : nullptr);
        st->lanczos_vecs = random ? lanczos_iter * cfg.GetNumLanczos() : lanczos_iter;
        st->t_lanczos = since(t0);

        std::size_t locked = 0, unconverged = nevex, iteration = 0;
        R lowerb = *std::max_element(ritzv_all, ritzv_all + unconverged);
        const R lambda = *std::min_element(ritzv_all, ritzv_all + nevex);     // never updated afterwards
        lowerb = lowerb * cfg.GetDecayingRate();
        if (tr) tr->add("bounds %.10e %.10e %.10e", (double)lambda, (double)lowerb, (double)upperb);

        while (unconverged > nex && iteration < cfg.GetMaxIter()) {
            R* ritzv = ritzv_all + locked;
            R* resid = resid_all + locked;
            R* residLast = residLast_all.data() + locked;
            std::size_t* degrees = degrees_all.data() + locked;

            bool all_small = true;
            for (std::size_t i = 0; i < unconverged; ++i)
                if (resid[i] > 5e-1) { all_small = false; break; }
            if (k->isSym() && all_small) lowerb = ritzv[unconverged - 1];
            if (lowerb > upperb) {
                std::fprintf(stderr, "chase_amd: lowerb > upperb, clamping\n");
                lowerb = upperb;
            }
            if (k->isSym())
                for (std::size_t i = 0; i < unconverged; ++i) residLast[i] = std::min(residLast[i], resid[i]);

            if (cfg.DoOptimization() && iteration != 0)
                deg = calc_degrees(k, unconverged, nex, upperb, lowerb, (R)tol, ritzv, resid, degrees, locked);

            t0 = clock::now();
            if (tr) tr->add("filter it=%zu unconverged=%zu deg=%zu", iteration, unconverged, deg);
            const std::size_t Av = filter(k, N, unconverged, deg, degrees, lambda, lowerb, upperb, tr);
            st->filtered_vecs += Av;
            st->t_filter += since(t0);

            // condition-number estimate of the filtered block steers the CholQR variant
            const R cc = (upperb + lowerb) / 2, ee = (upperb - lowerb) / 2;
            const R t_1 = (k->GetRitzv()[0] - cc) / ee, t_k = (ritzv[0] - cc) / ee;
            const R rho_1 = std::max(std::abs(t_1 - std::sqrt(t_1 * t_1 - 1)), std::abs(t_1 + std::sqrt(t_1 * t_1 - 1)));
            const R rho_k = std::max(std::abs(t_k - std::sqrt(t_k * t_k - 1)), std::abs(t_k + std::sqrt(t_k * t_k - 1)));
            const std::size_t dmax = *std::max_element(degrees, degrees + (nevex - locked));
            const R cond = std::pow(rho_k, (R)degrees[0]) * std::pow(rho_1, (R)(dmax - degrees[0]));

            t0 = clock::now();
            if (tr) tr->add("QR %zu %.6e", locked, (double)cond);
            k->QR(locked, cond);
            st->t_qr += since(t0);

            t0 = clock::now();
            if (tr) tr->add("RR %zu", unconverged);
            k->RR(ritzv, unconverged);
            st->t_rr += since(t0);

            t0 = clock::now();
            if (tr) tr->add("Resd %zu", locked);
            k->Resd(ritzv, resid, locked);
            st->t_resid += since(t0);

            const std::size_t new_converged =
                locking(k, unconverged - nex, (R)tol, ritzv, resid, residLast, &early, locked);
            if (tr) tr->add("Lock %zu", new_converged);
            k->Lock(new_converged);

            st->iter_unconverged.push_back(unconverged);
            st->iter_filtered.push_back(Av);
            st->iter_maxdeg.push_back(deg);
            locked += new_converged;
            unconverged -= new_converged;
            ++iteration;
            st->iterations = iteration; st->locked = locked;
            if (tr && tr->iter_hook && tr->iter_hook(tr->iter_user, iteration - 1, Av, locked, unconverged)) break;
        }

        // final ordering of the nev wanted pairs by eigenvalue: follow the cycles of the sorting permutation
        std::vector<std::size_t> perm(nev);
        std::iota(perm.begin(), perm.end(), 0);
        std::sort(perm.begin(), perm.end(), [&](std::size_t a, std::size_t b) { return ritzv_all[a] < ritzv_all[b]; });
        std::vector<bool> seen(nev, false);
        for (std::size_t i = 0; i < nev; ++i) {
            if (seen[i] || perm[i] == i) continue;
            std::vector<std::size_t> cyc;
            for (std::size_t cur = i; !seen[cur]; cur = perm[cur]) { seen[cur] = true; cyc.push_back(cur); }
            const R r0 = ritzv_all[cyc[0]], s0 = resid_all[cyc[0]];
            for (std::size_t q = 0; q + 1 < cyc.size(); ++q) {
                ritzv_all[cyc[q]] = ritzv_all[cyc[q + 1]];
                resid_all[cyc[q]] = resid_all[cyc[q + 1]];
            }
            ritzv_all[cyc.back()] = r0;
            resid_all[cyc.back()] = s0;
            for (std::size_t q = 0; q + 1 < cyc.size(); ++q) k->Swap(cyc[q], cyc[q + 1]);
        }
        k->set_early_locked_residuals(early);
        k->End();

        st->iterations = iteration;
        st->locked = locked;
        st->lowerb = lowerb; st->upperb = upperb; st->lambda = lambda;
        st->t_all = since(t_all0);
    }
};

} // namespace chase_amd
