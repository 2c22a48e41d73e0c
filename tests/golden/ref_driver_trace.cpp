// ref_driver_trace.cpp — TEST INFRASTRUCTURE, build container only (needs /root/reference).
//
// Runs the REFERENCE's own driver — chase::Solve = Algorithm<T>::solve, algorithm/algorithm.inc:1376-1788, compiled as it
// stands from /root/reference (the driver half of ChASE is header-only and BLAS-free) — on the naive CPU kernel of
// tests/cpu_mock_kernel.hpp deriving from the reference's chase::ChaseBase<double> (algorithm/interface.hpp:46-434), and
// prints iteration count, filtered-vector count, eigenpairs and the complete sequence of virtual calls with their scalar
// arguments.  tests/golden/make_driver_traces.sh commits the output as tests/golden/driver_trace_*.txt; no reference
// source is copied.
//   usage: ref_driver_trace N nev nex deg opt perturb [seq]   (seq = 1: second solve of the diagonally perturbed matrix from the
//          first solve's vectors, SetApprox(true))
#include "algorithm/algorithm.hpp"
#include "../cpu_mock_kernel.hpp"

int main(int argc, char** argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: %s N nev nex deg opt perturb\n", argv[0]); return 2; }
    const size_t N = std::atoi(argv[1]), nev = std::atoi(argv[2]), nex = std::atoi(argv[3]);
    const int deg = std::atoi(argv[4]), opt = std::atoi(argv[5]);
    const double perturb = std::atof(argv[6]);
    CpuMock<chase::ChaseBase<double>, chase::ChaseConfig<double>> k(N, nev, nex, clement_matrix(N, perturb));
    k.GetConfig().SetDeg(deg);
    k.GetConfig().SetOpt(opt != 0);
    chase::Solve(&k);
    if (argc > 7 && std::atoi(argv[7]) == 1) {
        k.perturb_diagonal(1e-3);
        k.GetConfig().SetApprox(true);
        chase::Solve(&k);
    }
    print_run(k, nev);
    return 0;
}
