// hello_world.cpp — C++ caller of the Impl classes, the way a ChASE application uses them (cf. the reference's
// examples/1_hello_world): construct the Impl on user buffers, set the configuration, run the solver driver.  Compiles
// with a plain host compiler (the Impl headers only need the C ABI):
//   g++ -O2 -std=c++17 -Iinclude -Ichase_amd/host examples/hello_world.cpp -Lchase_amd/lib -lchase_hip -Wl,-rpath,$PWD/chase_amd/lib
// Inside a ChASE checkout replace chase_amd::ChaseBase / ChaseConfig / Algorithm::solve by chase::ChaseBase /
// chase::ChaseConfig / chase::Solve.
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
#include "chase_hip.h"
#include "algorithm.hpp"
#include "chase_hip_impl.hpp"

using T = std::complex<double>;

int main()
{
    const std::size_t N = 1200, nev = 80, nex = 60;          // the reference example's problem
    std::vector<T> H(N * N, T(0)), V(N * (nev + nex));
    std::vector<double> Lambda(nev + nex);
    for (std::size_t i = 0; i + 1 < N; ++i) {
        const double v = std::sqrt((double)i * (double)(N + 1 - i));
        H[i + 1 + N * i] = v; H[i + N * (i + 1)] = v;
    }
    chase_hip_ctx* ctx = nullptr;
    if (chase_hip_ctx_create(&ctx, 0, nullptr) != 0) { std::fprintf(stderr, "%s\n", chase_hip_last_error()); return 2; }
    int rc = 1;
    {
        chase_amd::ChaseHip<T> single(ctx, N, nev, nex, H.data(), N, V.data(), N, Lambda.data());
        auto& config = single.GetConfig();
        config.SetTol(1e-10); config.SetDeg(20); config.SetOpt(true);
        chase_amd::SolveStats st;
        chase_amd::Algorithm<T, chase_amd::ChaseBase<T>>::solve(&single, &st, nullptr);
        std::printf("iterations %zu, filtered vectors %zu, lambda[0..2] = %.6f %.6f %.6f, resid[0] = %.3e\n",
                    (std::size_t)st.iterations, (std::size_t)st.filtered_vecs, Lambda[0], Lambda[1], Lambda[2],
                    single.GetResid()[0]);
        // the reference binary reports 5 iterations / 12 664 filtered vectors for this problem
        const bool ok = st.iterations == 5 && st.filtered_vecs == 12664 && std::abs(Lambda[0] + 1200.0) < 1e-8;
        if (ok) { std::printf("HELLO_WORLD_OK\n"); rc = 0; }
    }
    chase_hip_ctx_destroy(ctx);
    return rc;
}
