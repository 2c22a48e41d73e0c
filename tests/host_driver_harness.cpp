// host_driver_harness.cpp — TEST INFRASTRUCTURE, not part of the product.
//
// Runs the product's host-side solver driver (chase_amd/host/algorithm.hpp: solve / filter / calc_degrees / locking /
// lanczos — the restatement of ChASE's algorithm/algorithm.inc) WITHOUT a GPU, on the naive CPU kernel of
// tests/cpu_mock_kernel.hpp.  tests/test_host_driver.py compiles it with g++ and compares (a) with the Python oracle on the
// same matrix and (b) with tests/golden/driver_trace_*.txt, the runs of the REFERENCE's own driver on the same kernel.
// The mock is never linked into libchase_hip.so.
//   usage: host_driver_harness N nev nex deg opt [perturb [seq]]   (seq = 1: a second solve of the diagonally perturbed
//          matrix from the first solve's vectors, approximate mode)
#include "../chase_amd/host/algorithm.hpp"
#include "cpu_mock_kernel.hpp"

using namespace chase_amd;

int main(int argc, char** argv)
{
    const size_t N = argc > 1 ? std::atoi(argv[1]) : 96, nev = argc > 2 ? std::atoi(argv[2]) : 8,
                 nex = argc > 3 ? std::atoi(argv[3]) : 6;
    const int deg = argc > 4 ? std::atoi(argv[4]) : 10, opt = argc > 5 ? std::atoi(argv[5]) : 1;
    const double perturb = argc > 6 ? std::atof(argv[6]) : 0.0;
    CpuMock<ChaseBase<double>, ChaseConfig<double>> k(N, nev, nex, clement_matrix(N, perturb));
    k.GetConfig().SetDeg(deg);
    k.GetConfig().SetOpt(opt != 0);
    SolveStats st;
    CallTrace tr;
    tr.enabled = true;
    Algorithm<double, ChaseBase<double>>::solve(&k, &st, &tr);
    if (argc > 7 && std::atoi(argv[7]) == 1) {
        k.perturb_diagonal(1e-3);
        k.GetConfig().SetApprox(true);
        SolveStats st2;
        Algorithm<double, ChaseBase<double>>::solve(&k, &st2, &tr);
        st.iterations += st2.iterations; st.filtered_vecs += st2.filtered_vecs; st.locked = st2.locked;
    }
    std::printf("locked %zu\nstats_iterations %zu\nstats_filtered_vecs %zu\n", (size_t)st.locked, (size_t)st.iterations,
                (size_t)st.filtered_vecs);
    print_run(k, nev);
    for (const auto& l : tr.lines) std::printf("trace %s\n", l.c_str());
    return 0;
}
