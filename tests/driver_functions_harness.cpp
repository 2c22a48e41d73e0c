// driver_functions_harness.cpp — TEST INFRASTRUCTURE, not part of the product.
//
// The product's solver driver (chase_amd/host/algorithm.hpp) on the function-level scenarios of
// tests/driver_function_scenarios.hpp; its output must equal tests/golden/driver_functions.txt, the output of the
// REFERENCE's routines on the same inputs (tests/golden/ref_driver_functions.cpp).
#include "../chase_amd/host/algorithm.hpp"
#include "driver_function_scenarios.hpp"

struct OwnCalls {
    using Kernel = scen::ScriptKernel<chase_amd::ChaseBase<double>, chase_amd::ChaseConfig<double>>;
    using A = chase_amd::Algorithm<double, chase_amd::ChaseBase<double>>;
    static std::size_t calc_degrees(Kernel* k, std::size_t, std::size_t unc, std::size_t nex, double ub, double lb, double tol,
                                    double* ritzv, double* resid, std::size_t* deg, std::size_t locked)
    {
        return A::calc_degrees(k, unc, nex, ub, lb, tol, ritzv, resid, deg, locked);
    }
    static std::size_t locking(Kernel* k, std::size_t, std::size_t cand, double tol, double* ritzv, double* resid, double* residLast,
                               std::vector<double>* early, std::size_t*, std::size_t locked)
    {
        return A::locking(k, cand, tol, ritzv, resid, residLast, early, locked);
    }
    static std::size_t filter(Kernel* k, std::size_t n, std::size_t unp, std::size_t deg, std::size_t* degrees, double l1, double lo,
                              double up)
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
    static std::size_t calc_degrees_pseudo_H2(Kernel* k, std::size_t, std::size_t unc, std::size_t nex, double ub, double lb, double tol,
                                              double* ritzv, double* resid, double* residLast, std::size_t* deg, std::size_t locked)
    {
        return A::calc_degrees_pseudo_H2(k, unc, nex, ub, lb, tol, ritzv, resid, residLast, deg, locked);
    }
    static std::size_t locking_pseudo_v3(Kernel* k, std::size_t, std::size_t unc, std::size_t nex, double tol, std::size_t* index,
                                         double* ritzv, double* resid, double* residLast, std::vector<double>* early, std::size_t*,
                                         std::size_t locked, std::size_t iteration, std::size_t)
    {
        return A::locking_pseudo_v3(k, unc, nex, tol, index, ritzv, resid, residLast, early, locked, iteration);
    }
    static std::size_t filter_H2(Kernel* k, std::size_t, std::size_t unc, std::size_t* degrees, double l1, double lo, double up)
    {
        return A::filter_H2(k, unc, degrees, l1, lo, up);
    }
    static std::size_t lanczos_for_H2(Kernel* k, int N, int nv, int m, int nevex, double* ub, double* ritzv)
    {
        return A::lanczos_for_H2(k, N, nv, m, nevex, ub, ritzv);
    }
    static void solve(Kernel* k) { A::solve(k); }
    static void solve_pseudo(Kernel* k) { A::solve_pseudo(k); }
};

int main()
{
    scen::run_all<OwnCalls>();
    return 0;
}
