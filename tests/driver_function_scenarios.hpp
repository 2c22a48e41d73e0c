// driver_function_scenarios.hpp — TEST INFRASTRUCTURE, not part of the product.
//
// Function-level known answers for the solver driver: the scalar routines of algorithm/algorithm.inc are driven one by one
// on seeded synthetic inputs through a SCRIPTED kernel (it only logs the virtual calls it receives; its Lanczos overloads
// return scripted Ritz data), including the branches whole solves rarely reach (early locking of stagnating pairs, ties in
// the degree sort, cluster factors, the DoS scan tripping at either end, swapped filter bounds).  Two programs share this
// header and must print identical text:
//   tests/golden/ref_driver_functions.cpp   the REFERENCE's chase::Algorithm<double> (build container; output committed as
//                                           tests/golden/driver_functions.txt, also built into oracle/_ref/)
//   tests/driver_functions_harness.cpp      the product's chase_amd::Algorithm<double, ...>
// tests/test_reference_driver_functions.py additionally runs the Python oracle's restatements on the inputs printed here.
// `Calls` adapts the two drivers' slightly different signatures (the reference passes N / degrees to some routines).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <string>
#include <vector>

namespace scen {

struct Lcg {                                   // deterministic uniform (0,1) numbers, identical in both programs
    std::uint64_t s;
    explicit Lcg(std::uint64_t seed) : s(seed * 2862933555777941757ull + 3037000493ull) {}
    double u()
    {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return ((double)(s >> 11) + 0.5) / 9007199254740992.0;
    }
};

inline void put(const char* tag, const char* name, const std::vector<double>& v)
{
    std::printf("%s %s %zu", tag, name, v.size());
    for (double x : v) std::printf(" %.17g", x);
    std::printf("\n");
}
inline void put(const char* tag, const char* name, const std::vector<std::size_t>& v)
{
    std::printf("%s %s %zu", tag, name, v.size());
    for (std::size_t x : v) std::printf(" %zu", x);
    std::printf("\n");
}
inline void put1(const char* tag, const char* name, double x) { std::printf("%s %s 1 %.17g\n", tag, name, x); }

// Kernel that records the virtual calls of the driver routine under test.  Lanczos(M, numvec, ...) hands out scripted data.
template <class BaseT, class ConfigT>
class ScriptKernel : public BaseT {
public:
    ScriptKernel(std::size_t N, std::size_t nev, std::size_t nex, bool pseudo)
        : N_(N), nev_(nev), nex_(nex), pseudo_(pseudo), cfg_(N, nev, nex), ritzv_(2 * (nev + nex)), resid_(2 * (nev + nex)) {}
    std::vector<std::string> calls;
    std::vector<double> theta, tau, ritzV;     // script of the multi-vector Lanczos
    double upperb_script = 0;
    void flush()
    {
        for (const auto& c : calls) std::printf("C %s\n", c.c_str());
        calls.clear();
    }
    void log(const char* fmt, ...) __attribute__((format(printf, 2, 3)))
    {
        char buf[256];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        calls.emplace_back(buf);
    }
    void Shift(double c, bool u = false) override { log("Shift %.17g %d", c, (int)u); }
    void HEMM(std::size_t b, double al, double be, std::size_t ol, std::size_t orr = 0) override
    {
        log("HEMM %zu %.17g %.17g %zu %zu", b, al, be, ol, orr);
    }
    void HEMM_H2(std::size_t b, double al, double be, double ga, std::size_t ol, std::size_t orr = 0) override
    {
        log("HEMM_H2 %zu %.17g %.17g %.17g %zu %zu", b, al, be, ga, ol, orr);
    }
    void ApplyKconjugate(std::size_t b) override { log("ApplyKconjugate %zu", b); }
    void FilterPhaseStart() override { log("FilterPhaseStart"); }
    void FilterPhaseEnd() override { log("FilterPhaseEnd"); }
    void QR(std::size_t f, double c) override { log("QR %zu %.17g", f, c); }
    // whole-solve scenarios: Ritz values and residuals come from a script that ignores the vectors - column position j of
    // the block yields eig[j] (slightly moving with the iteration) and a residual that decays geometrically at its own rate.
    // The driver under test only ever sees these numbers, so two drivers fed the same script must make the same decisions.
    std::vector<double> eig, r0, decay;
    std::size_t locked_ = 0, it_ = 0;
    void RR(double* ritzv, std::size_t b) override
    {
        log("RR %zu", b);
        if (eig.empty()) return;
        for (std::size_t i = 0; i < b; ++i) ritzv[i] = eig[locked_ + i] * (1.0 + 1e-3 / (double)(it_ + 1));
        if (pseudo_) for (std::size_t i = 0; i < b; ++i) ritzv[b + i] = -ritzv[i];         // the K-conjugate partners
        ++it_;
    }
    void Sort(double*, double*, double*) override {}
    void Resd(double*, double* resid, std::size_t f) override
    {
        log("Resd %zu", f);
        if (eig.empty()) return;
        const std::size_t sub = nev_ + nex_ - locked_;
        for (std::size_t i = 0; i < sub; ++i) resid[i] = r0[locked_ + i] * std::pow(decay[locked_ + i], (double)it_);
    }
    void Lanczos(std::size_t m, double* ub) override { log("Lanczos1 %zu", m); *ub = upperb_script; }
    void Lanczos(std::size_t M, std::size_t nv, double* ub, double* rv, double* Tau, double* rV) override
    {
        log("Lanczos %zu %zu", M, nv);
        *ub = upperb_script;
        std::copy(theta.begin(), theta.begin() + M * nv, rv);
        std::copy(tau.begin(), tau.begin() + M * nv, Tau);
        std::copy(ritzV.begin(), ritzV.begin() + M * M, rV);
    }
    void LanczosDos(std::size_t idx, std::size_t m, double*) override { log("LanczosDos %zu %zu", idx, m); }
    void Swap(std::size_t i, std::size_t j) override { log("Swap %zu %zu", i, j); }
    void Lock(std::size_t k) override { log("Lock %zu", k); locked_ += k; }
    bool checkSymmetryEasy() override { return !pseudo_; }
    bool isSym() override { return !pseudo_; }
    bool checkPseudoHermicityEasy() override { return pseudo_; }
    bool isPseudoHerm() override { return pseudo_; }
    void symOrHermMatrix(char) override {}
    void Start() override { log("Start"); locked_ = 0; it_ = 0; }
    void End() override { log("End"); }
    void initVecs(bool r) override { log("initVecs %d", (int)r); }
    std::size_t GetN() const override { return N_; }
    std::size_t GetNev() override { return nev_; }
    std::size_t GetNex() override { return nex_; }
    std::size_t GetLanczosIter() override { return 0; }
    std::size_t GetNumLanczos() override { return 0; }
    std::size_t GetRitzvBlockSize() const override { return (pseudo_ ? 2 : 1) * (nev_ + nex_); }
    double* GetRitzv() override { return ritzv_.data(); }
    double* GetResid() override { return resid_.data(); }
    ConfigT& GetConfig() override { return cfg_; }
    int get_nprocs() override { return 1; }
    int get_rank() override { return 0; }

private:
    std::size_t N_, nev_, nex_;
    bool pseudo_;
    ConfigT cfg_;
    std::vector<double> ritzv_, resid_;
};

// sorted "Ritz values" in [lo, hi] with a few clusters, and residuals spread over many decades around tol
inline void make_pairs(Lcg& g, std::size_t n, double lo, double hi, double tol, std::vector<double>& ritzv, std::vector<double>& resid,
                       bool clusters)
{
    ritzv.resize(n); resid.resize(n);
    for (std::size_t i = 0; i < n; ++i) ritzv[i] = lo + (hi - lo) * g.u();
    std::sort(ritzv.begin(), ritzv.end());
    if (clusters)
        for (std::size_t i = 3; i + 1 < n; i += 5) ritzv[i + 1] = ritzv[i] + (hi - lo) * 1e-8 * g.u();
    for (std::size_t i = 0; i < n; ++i) resid[i] = tol * std::pow(10.0, -1.5 + 9.0 * g.u());
}

template <class Calls>
void run_all()
{
    using K = typename Calls::Kernel;
    const double tol = 1e-10;
    // ---- calc_degrees (Hermitian): algorithm.inc:136-193 -------------------------------------------------------------
    for (int sc = 0; sc < 4; ++sc) {
        Lcg g(100 + sc);
        const std::size_t nev = 10 + 3 * sc, nex = 4 + sc, unconverged = nev + nex - (sc == 2 ? 3 : 0), locked = (sc == 2 ? 3 : 0);
        K k(400, nev, nex, false);
        if (sc == 3) { k.GetConfig().SetDegExtra(5); k.GetConfig().SetMaxDeg(24); }
        std::vector<double> ritzv, resid;
        make_pairs(g, unconverged, -3.0, 1.0, tol, ritzv, resid, sc == 1);
        if (sc == 1) for (std::size_t i = 0; i + 1 < unconverged; i += 2) resid[i + 1] = resid[i];      // ties in the degree sort
        const double lowerb = 1.5, upperb = 9.0;
        std::vector<std::size_t> degrees(unconverged, 20);
        std::printf("S calc_degrees %d\n", sc);
        put("I", "ritzv", ritzv); put("I", "resid", resid);
        std::printf("I params 9 %zu %zu %.17g %.17g %.17g %zu %zu %zu %d\n", unconverged, nex, upperb, lowerb, tol, locked,
                    (std::size_t)k.GetConfig().GetDegExtra(), (std::size_t)k.GetConfig().GetMaxDeg(), 0);
        const std::size_t ret = Calls::calc_degrees(&k, 400, unconverged, nex, upperb, lowerb, tol, ritzv.data(), resid.data(),
                                                    degrees.data(), locked);
        put("O", "degrees", degrees); put("O", "ritzv", ritzv); put("O", "resid", resid);
        std::printf("O ret 1 %zu\n", ret);
        k.flush();
    }
    // ---- locking (Hermitian): algorithm.inc:519-578 ----------------------------------------------------------------------
    for (int sc = 0; sc < 4; ++sc) {
        Lcg g(200 + sc);
        const std::size_t nev = 12, nex = 5, candidates = nev, locked = (sc == 3 ? 4 : 0);
        K k(400, nev, nex, false);
        std::vector<double> ritzv, resid, residLast(candidates);
        make_pairs(g, candidates, -5.0, -1.0, tol, ritzv, resid, false);
        // unsorted Ritz values (the routine visits them in ascending order through an index array)
        for (std::size_t i = 0; i + 1 < candidates; i += 3) std::swap(ritzv[i], ritzv[i + 1]);
        for (std::size_t i = 0; i < candidates; ++i) {
            residLast[i] = resid[i] * (0.2 + 1.6 * g.u());         // some pairs stagnate (resid >= residLast) -> early lock
            if (sc == 1) resid[i] = tol * (i % 2 ? 0.5 : 30.0);     // half converged, half stagnating inside 100 tol
            if (sc == 2) resid[i] = tol * 1e3;                      // nothing lockable
        }
        std::vector<double> early;
        std::vector<std::size_t> degrees(candidates + nex, 20);
        std::printf("S locking %d\n", sc);
        put("I", "ritzv", ritzv); put("I", "resid", resid); put("I", "residLast", residLast);
        std::printf("I params 3 %zu %.17g %zu\n", candidates, tol, locked);
        const std::size_t ret = Calls::locking(&k, 400, candidates, tol, ritzv.data(), resid.data(), residLast.data(), &early,
                                               degrees.data(), locked);
        put("O", "ritzv", ritzv); put("O", "resid", resid); put("O", "residLast", residLast); put("O", "early", early);
        std::printf("O ret 1 %zu\n", ret);
        k.flush();
    }
    // ---- filter (Hermitian): algorithm.inc:942-1009 ------------------------------------------------------------------------
    for (int sc = 0; sc < 3; ++sc) {
        const std::size_t nev = 9, nex = 4, n = 400, unprocessed = nev + nex;
        K k(n, nev, nex, false);
        std::vector<std::size_t> degrees(unprocessed);
        std::size_t deg = 0;
        for (std::size_t i = 0; i < unprocessed; ++i) { degrees[i] = sc == 0 ? 20 : 2 * (1 + i / 2 + (sc == 2 ? i / 3 : 0)); deg = std::max(deg, degrees[i]); }
        std::sort(degrees.begin(), degrees.end());
        std::printf("S filter %d\n", sc);
        put("I", "degrees", degrees);
        const double lambda_1 = -7.5, lower = -1.25 + sc, upper = 6.0;
        std::printf("I params 6 %zu %zu %zu %.17g %.17g %.17g\n", n, unprocessed, deg, lambda_1, lower, upper);
        const std::size_t ret = Calls::filter(&k, n, unprocessed, deg, degrees.data(), lambda_1, lower, upper);
        std::printf("O ret 1 %zu\n", ret);
        k.flush();
    }
    // ---- lanczos + DoS (Hermitian): algorithm.inc:1067-1214 -----------------------------------------------------------------
    for (int sc = 0; sc < 4; ++sc) {
        Lcg g(400 + sc);
        const int N = 500 + 100 * sc, numvec = 4, m = 12, nev = 20 + 10 * sc, nex = 10, nevex = nev + nex;
        K k((std::size_t)N, (std::size_t)nev, (std::size_t)nex, false);
        k.theta.resize((std::size_t)numvec * m); k.tau.resize((std::size_t)numvec * m); k.ritzV.resize((std::size_t)m * m);
        for (int r = 0; r < numvec; ++r) {
            std::vector<double> th(m), w(m);
            double ws = 0;
            for (int i = 0; i < m; ++i) { th[i] = -10.0 + 20.0 * g.u(); w[i] = 0.05 + g.u(); ws += w[i]; }
            if (sc == 3) for (int i = 0; i < m; ++i) th[i] = 5.0 + 1e-3 * i - r * 1e-5;       // narrow spectrum: the scan trips late
            // the LAST run must own the smallest Ritz value: otherwise the reference's extraction index becomes -1 and it
            // writes ritzv_[-1] (algorithm.inc:1161-1196; undefined behaviour the product's restatement guards against)
            if (r == numvec - 1 && sc != 3) th[0] = -10.5;
            std::sort(th.begin(), th.end());
            for (int i = 0; i < m; ++i) { k.theta[(std::size_t)r * m + i] = th[i]; k.tau[(std::size_t)r * m + i] = w[i] / ws; }
        }
        for (auto& x : k.ritzV) x = g.u() - 0.5;
        k.upperb_script = 11.0 + sc;
        std::vector<double> ritzv((std::size_t)nevex, 0.0);
        double upperb = 0;
        std::printf("S lanczos %d\n", sc);
        put("I", "theta", k.theta); put("I", "tau", k.tau); put("I", "ritzV", k.ritzV);
        std::printf("I params 6 %d %d %d %d %.17g %d\n", N, numvec, m, nevex, k.upperb_script, sc == 1 ? 0 : 1);
        const std::size_t ret = Calls::lanczos(&k, N, numvec, m, nevex, &upperb, sc != 1, ritzv.data());
        if (sc != 1) put("O", "ritzv", ritzv);
        put1("O", "upperb", upperb);
        std::printf("O ret 1 %zu\n", ret);
        k.flush();
    }
    // ---- pseudo-Hermitian: detect_eigenvalue_clusters :19-133, calc_degrees_pseudo_H2 :196-317 ------------------------------
    for (int sc = 0; sc < 4; ++sc) {
        Lcg g(500 + sc);
        const std::size_t nev = 14 + 2 * sc, nex = 6, unconverged = nev + nex - (sc == 3 ? 4 : 0), locked = (sc == 3 ? 4 : 0);
        K k(600, nev, nex, true);
        if (sc == 2) k.GetConfig().SetClusterAwareDegrees(false);
        std::vector<double> ritzv, resid, residLast(unconverged);
        make_pairs(g, unconverged, 0.8, 3.0, tol, ritzv, resid, sc != 2);
        for (std::size_t i = 0; i < unconverged; ++i) residLast[i] = resid[i] * (sc == 1 ? 1.02 : 0.3 + 2.0 * g.u());   // sc 1: stagnation bonus
        const double lowerb = sc == 0 ? 12.0 : 10.5, upperb = 95.0;
        std::vector<double> cf;
        std::printf("S clusters %d\n", sc);
        put("I", "ritzv", ritzv); put("I", "resid", resid);
        std::printf("I params 5 %zu %zu %.17g %.17g %.17g\n", unconverged, nex, upperb, lowerb, tol);
        Calls::detect_eigenvalue_clusters(ritzv.data(), resid.data(), tol, unconverged, nex, upperb, lowerb, cf);
        put("O", "factors", cf);
        std::vector<std::size_t> degrees(unconverged, 20);
        std::printf("S calc_degrees_pseudo_H2 %d\n", sc);
        put("I", "ritzv", ritzv); put("I", "resid", resid); put("I", "residLast", residLast);
        std::printf("I params 7 %zu %zu %.17g %.17g %.17g %zu %d\n", unconverged, nex, upperb, lowerb, tol, locked,
                    (int)k.GetConfig().UseClusterAwareDegrees());
        const std::size_t ret = Calls::calc_degrees_pseudo_H2(&k, 600, unconverged, nex, upperb, lowerb, tol, ritzv.data(),
                                                              resid.data(), residLast.data(), degrees.data(), locked);
        put("O", "degrees", degrees); put("O", "ritzv", ritzv); put("O", "resid", resid);
        std::printf("O ret 1 %zu\n", ret);
        k.flush();
    }
    // ---- locking_pseudo_v3: algorithm.inc:730-817 ---------------------------------------------------------------------------
    for (int sc = 0; sc < 4; ++sc) {
        Lcg g(600 + sc);
        const std::size_t nev = 10, nex = 4, unconverged = nev + nex, locked = (sc == 2 ? 2 : 0), iteration = (sc == 0 ? 1 : 5);
        K k(600, nev, nex, true);
        std::vector<double> ritzv, resid, residLast(2 * unconverged);
        make_pairs(g, 2 * unconverged, 0.5, 4.0, tol, ritzv, resid, false);
        for (std::size_t i = 0; i < 2 * unconverged; ++i) {
            residLast[i] = resid[i] * (0.3 + 1.5 * g.u());
            if (sc == 3) resid[i] = tol * (i % 3 == 0 ? 0.3 : (i % 3 == 1 ? 500.0 : 5000.0));
        }
        std::vector<std::size_t> index(2 * unconverged), degrees(2 * unconverged, 20);
        for (std::size_t i = 0; i < 2 * unconverged; ++i) index[i] = i;
        if (sc == 1) for (std::size_t i = 0; i + 1 < unconverged; i += 2) std::swap(index[i], index[i + 1]);
        std::vector<double> early;
        std::printf("S locking_pseudo_v3 %d\n", sc);
        put("I", "ritzv", ritzv); put("I", "resid", resid); put("I", "residLast", residLast); put("I", "index", index);
        std::printf("I params 5 %zu %zu %.17g %zu %zu\n", unconverged, nex, tol, locked, iteration);
        const std::size_t ret = Calls::locking_pseudo_v3(&k, 600, unconverged, nex, tol, index.data(), ritzv.data(), resid.data(),
                                                         residLast.data(), &early, degrees.data(), locked, iteration, nev);
        put("O", "ritzv", ritzv); put("O", "resid", resid); put("O", "residLast", residLast); put("O", "early", early);
        std::printf("O ret 1 %zu\n", ret);
        k.flush();
    }
    // ---- filter_H2: algorithm.inc:1012-1064 -----------------------------------------------------------------------------------
    for (int sc = 0; sc < 3; ++sc) {
        const std::size_t nev = 8, nex = 4, unconverged = nev + nex;
        K k(600, nev, nex, true);
        std::vector<std::size_t> degrees(unconverged);
        for (std::size_t i = 0; i < unconverged; ++i) degrees[i] = sc == 0 ? 12 : 2 * (2 + i / 2);
        std::printf("S filter_H2 %d\n", sc);
        put("I", "degrees", degrees);
        const double lambda_1 = 0.9, lower = sc == 2 ? 90.0 : 14.0, upper = sc == 2 ? 14.0 : 90.0;      // sc 2: swapped bounds
        std::printf("I params 4 %zu %.17g %.17g %.17g\n", unconverged, lambda_1, lower, upper);
        const std::size_t ret = Calls::filter_H2(&k, 600, unconverged, degrees.data(), lambda_1, lower, upper);
        std::printf("O ret 1 %zu\n", ret);
        k.flush();
    }
    // ---- lanczos_for_H2: algorithm.inc:1217-1373 --------------------------------------------------------------------------------
    for (int sc = 0; sc < 3; ++sc) {
        Lcg g(800 + sc);
        const int N = 600 + 200 * sc, numvec = 5, m = 10, nev = 20 + 15 * sc, nex = 10, nevex = nev + nex;
        K k((std::size_t)N, (std::size_t)nev, (std::size_t)nex, true);
        k.theta.resize((std::size_t)numvec * m); k.tau.resize((std::size_t)numvec * m); k.ritzV.resize((std::size_t)m * m);
        for (int r = 0; r < numvec; ++r) {
            // +/- pairs like a pseudo-Hermitian spectrum
            std::vector<double> th(m), w(m);
            double ws = 0;
            for (int i = 0; i < m / 2; ++i) { const double a = 1.0 + 9.0 * g.u(); th[i] = -a; th[m - 1 - i] = a * (1 + 1e-3 * g.u()); }
            for (int i = 0; i < m; ++i) { w[i] = 0.05 + g.u(); ws += w[i]; }
            std::sort(th.begin(), th.end());
            for (int i = 0; i < m; ++i) { k.theta[(std::size_t)r * m + i] = th[i]; k.tau[(std::size_t)r * m + i] = w[i] / ws; }
        }
        for (auto& x : k.ritzV) x = g.u() - 0.5;
        k.upperb_script = 10.5;
        std::vector<double> ritzv(2 * (std::size_t)nevex, 0.0);
        double upperb = 0;
        std::printf("S lanczos_for_H2 %d\n", sc);
        put("I", "theta", k.theta); put("I", "tau", k.tau); put("I", "ritzV", k.ritzV);
        std::printf("I params 6 %d %d %d %d %d %d\n", N, numvec, m, nevex, nev, nex);
        const std::size_t ret = Calls::lanczos_for_H2(&k, N, numvec, m, nevex, &upperb, ritzv.data());
        ritzv.resize((std::size_t)nevex);
        put("O", "ritzv", ritzv);
        put1("O", "upperb", upperb);
        std::printf("O ret 1 %zu\n", ret);
        k.flush();
    }
    // ---- whole solves on scripted Ritz values / residuals: solve :1376-1788 and solve_pseudo :1834-2220 ---------------------
    // (control flow only: bounds, degree optimisation, condition estimates, locking, K-conjugation calls, final ordering)
    for (int pseudo = 0; pseudo < 2; ++pseudo)
        for (int sc = 0; sc < 3; ++sc) {
            Lcg g(900 + 10 * pseudo + sc);
            const std::size_t nev = 12 + 4 * sc, nex = 5 + sc, nevex = nev + nex;
            const int N = 2000, numvec = 4, m = (int)std::min<std::size_t>(nevex, 24) / 2 * 2;
            K k((std::size_t)N, nev, nex, pseudo != 0);
            if (sc == 1) k.GetConfig().SetOpt(false);
            if (sc == 2) { k.GetConfig().SetDeg(12); k.GetConfig().SetMaxIter(9); }
            k.theta.resize((std::size_t)numvec * m); k.tau.resize((std::size_t)numvec * m); k.ritzV.resize((std::size_t)m * m);
            for (int r = 0; r < numvec; ++r) {
                std::vector<double> th(m), w(m);
                double ws = 0;
                if (pseudo) for (int i = 0; i < m / 2; ++i) { const double a = 1.5 + 8.5 * g.u(); th[i] = -a; th[m - 1 - i] = a; }
                else { for (int i = 0; i < m; ++i) th[i] = -10.0 + 20.0 * g.u(); if (r == numvec - 1) th[0] = -10.5; }
                for (int i = 0; i < m; ++i) { w[i] = 0.05 + g.u(); ws += w[i]; }
                std::sort(th.begin(), th.end());
                for (int i = 0; i < m; ++i) { k.theta[(std::size_t)r * m + i] = th[i]; k.tau[(std::size_t)r * m + i] = w[i] / ws; }
            }
            for (auto& x : k.ritzV) x = g.u() - 0.5;
            k.upperb_script = 11.0;
            // wanted eigenvalues well outside the filter's damped interval; residuals shrink at 0.02 .. 0.2 per iteration
            k.eig.resize(nevex); k.r0.resize(nevex); k.decay.resize(nevex);
            for (std::size_t i = 0; i < nevex; ++i) k.eig[i] = pseudo ? 0.3 + 0.6 * g.u() : -30.0 + 10.0 * g.u();
            std::sort(k.eig.begin(), k.eig.end());
            for (std::size_t i = 0; i < nevex; ++i) { k.r0[i] = 0.05 + 0.3 * g.u(); k.decay[i] = 0.02 + 0.18 * g.u(); }
            if (sc == 2) for (std::size_t i = 0; i < nevex; i += 4) k.decay[i] = 0.9;        // stragglers: maxIter / early-lock paths
            std::printf("S %s %d\n", pseudo ? "solve_pseudo" : "solve", sc);
            put("I", "theta", k.theta); put("I", "tau", k.tau); put("I", "ritzV", k.ritzV);
            put("I", "eig", k.eig); put("I", "r0", k.r0); put("I", "decay", k.decay);
            std::printf("I params 8 %d %d %d %zu %zu %d %zu %zu\n", N, numvec, m, nev, nex, (int)k.GetConfig().DoOptimization(),
                        (std::size_t)k.GetConfig().GetDeg(), (std::size_t)k.GetConfig().GetMaxIter());
            k.GetConfig().SetNumLanczos((std::size_t)numvec);
            k.GetConfig().SetLanczosIter((std::size_t)m);
            if (pseudo) Calls::solve_pseudo(&k); else Calls::solve(&k);
            std::vector<double> rv(k.GetRitzv(), k.GetRitzv() + nevex), rs(k.GetResid(), k.GetResid() + nevex);
            put("O", "ritzv", rv); put("O", "resid", rs);
            std::printf("O locked 1 %zu\n", k.locked_);
            k.flush();
        }
}

} // namespace scen
