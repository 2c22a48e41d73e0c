// ref_driver_functions.cpp — TEST INFRASTRUCTURE, build container only (needs /root/reference).
//
// Drives the scalar routines of the REFERENCE's solver driver (chase::Algorithm<double>, algorithm/algorithm.inc, compiled
// as it stands from /root/reference) one by one on the seeded inputs of tests/driver_function_scenarios.hpp and prints
// inputs, outputs and the virtual calls each routine made.  Output committed as tests/golden/driver_functions.txt by
// tests/golden/make_driver_traces.sh; also built into oracle/_ref/ (oracle/Makefile).  No reference source is copied.
#include "algorithm/algorithm.hpp"
#include "../driver_function_scenarios.hpp"

struct RefCalls {
    using Kernel = scen::ScriptKernel<chase::ChaseBase<double>, chase::ChaseConfig<double>>;
    using A = chase::Algorithm<double>;
    static std::size_t calc_degrees(Kernel* k, std::size_t N, std::size_t unc, std::size_t nex, double ub, double lb, double tol,
                                    double* ritzv, double* resid, std::size_t* deg, std::size_t locked)
    {
        std::vector<double> residLast(unc, 0.0);             // not read by the Hermitian routine
        return A::calc_degrees(k, N, unc, nex, ub, lb, tol, ritzv, resid, residLast.data(), deg, locked);
    }
    static std::size_t locking(Kernel* k, std::size_t N, std::size_t cand, double tol, double* ritzv, double* resid,
                               double* residLast, std::vector<double>* early, std::size_t* deg, std::size_t locked)
    {
        return A::locking(k, N, cand, tol, ritzv, resid, residLast, early, deg, locked);
    }
    static std::size_t filter(Kernel* k, std::size_t n, std::size_t unp, std::size_t deg, std::size_t* degrees, double l1,
                              double lo, double up)
    {
        return A::filter(k, n, unp, deg, degrees, l1, lo, up);
    }
    static std::size_t lanczos(Kernel* k, int N, int nv, int m, int nevex, double* ub, bool mode, double* ritzv)
    {
        return A::lanczos(k, N, nv, m, nevex, ub, mode, ritzv);
    }
    static void detect_eigenvalue_clusters(double* ritzv, double* resid, double tol, std::size_t unc, std::size_t nex, double ub,
                                           double lb, std::vector<double>& f)
    {
        A::detect_eigenvalue_clusters(ritzv, resid, tol, unc, nex, ub, lb, f);
    }
    static std::size_t calc_degrees_pseudo_H2(Kernel* k, std::size_t N, std::size_t unc, std::size_t nex, double ub, double lb,
                                              double tol, double* ritzv, double* resid, double* residLast, std::size_t* deg,
                                              std::size_t locked)
    {
        return A::calc_degrees_pseudo_H2(k, N, unc, nex, ub, lb, tol, ritzv, resid, residLast, deg, locked);
    }
    static std::size_t locking_pseudo_v3(Kernel* k, std::size_t N, std::size_t unc, std::size_t nex, double tol, std::size_t* index,
                                         double* ritzv, double* resid, double* residLast, std::vector<double>* early,
                                         std::size_t* deg, std::size_t locked, std::size_t iteration, std::size_t nev)
    {
        return A::locking_pseudo_v3(k, N, unc, nex, tol, index, ritzv, resid, residLast, early, deg, locked, iteration, nev);
    }
    static std::size_t filter_H2(Kernel* k, std::size_t n, std::size_t unc, std::size_t* degrees, double l1, double lo, double up)
    {
        return A::filter_H2(k, n, unc, degrees, l1, lo, up);
    }
    static std::size_t lanczos_for_H2(Kernel* k, int N, int nv, int m, int nevex, double* ub, double* ritzv)
    {
        return A::lanczos_for_H2(k, N, nv, m, nevex, ub, true, ritzv);
    }
    static void solve(Kernel* k) { A::solve(k); }
    static void solve_pseudo(Kernel* k) { A::solve_pseudo(k); }
};

int main()
{
    scen::run_all<RefCalls>();
    return 0;
}
