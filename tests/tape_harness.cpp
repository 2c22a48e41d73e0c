// tape_harness.cpp — TEST INFRASTRUCTURE, not part of the product.
//
// The scalar tape of chase_amd/host/tape.hpp without a GPU, on the naive CPU kernel of tests/cpu_mock_kernel.hpp:
//   1. solve problem A (Clement-type matrix) through TapeKernel in RECORD mode: kernel-side call list A, tape;
//   2. solve a DIFFERENT problem B (other matrix: its own numbers would steer the driver elsewhere) through TapeKernel in
//      REPLAY mode on A's tape: the driver must issue A's call list, call for call and scalar for scalar;
//   3. B once more without the tape (the control: its own call list differs from A's);
//   4. replay on a truncated tape: must fail with "tape:" instead of running on.
//   usage: tape_harness N nev nex deg
//          tape_harness pseudo       the same three runs of chase::Solve_pseudo on the scripted kernel of
//                                    tests/driver_function_scenarios.hpp (script A recorded, script B replayed / on its own)
#include <cstring>
#include "../chase_amd/host/algorithm.hpp"
#include "../chase_amd/host/tape.hpp"
#include "cpu_mock_kernel.hpp"
#include "driver_function_scenarios.hpp"

using namespace chase_amd;
using Mock = CpuMock<ChaseBase<double>, ChaseConfig<double>>;

static std::vector<double> other_matrix(size_t N)
{
    std::vector<double> H = clement_matrix(N, 1e-2);
    for (auto& x : H) x *= 0.37;
    for (size_t i = 0; i < N; ++i) H[i + i * N] += 0.05 * (double)((i * 7) % 11);
    return H;
}

// a scripted pseudo-Hermitian kernel: what it tells the driver depends on the seed only
using Script = scen::ScriptKernel<ChaseBase<double>, ChaseConfig<double>>;
static void script(Script& k, unsigned seed, std::size_t nevex, int numvec, int m)
{
    scen::Lcg g(seed);
    k.theta.resize((std::size_t)numvec * m); k.tau.resize((std::size_t)numvec * m); k.ritzV.resize((std::size_t)m * m);
    for (int r = 0; r < numvec; ++r) {
        std::vector<double> th(m), w(m);
        double ws = 0;
        for (int i = 0; i < m / 2; ++i) { const double a = 1.5 + 8.5 * g.u(); th[i] = -a; th[m - 1 - i] = a; }
        for (int i = 0; i < m; ++i) { w[i] = 0.05 + g.u(); ws += w[i]; }
        std::sort(th.begin(), th.end());
        for (int i = 0; i < m; ++i) { k.theta[(std::size_t)r * m + i] = th[i]; k.tau[(std::size_t)r * m + i] = w[i] / ws; }
    }
    for (auto& x : k.ritzV) x = g.u() - 0.5;
    k.upperb_script = 11.0;
    k.eig.resize(nevex); k.r0.resize(nevex); k.decay.resize(nevex);
    for (std::size_t i = 0; i < nevex; ++i) k.eig[i] = 0.3 + 0.6 * g.u();
    std::sort(k.eig.begin(), k.eig.end());
    for (std::size_t i = 0; i < nevex; ++i) { k.r0[i] = 0.05 + 0.3 * g.u(); k.decay[i] = 0.02 + 0.18 * g.u(); }
    k.GetConfig().SetNumLanczos((std::size_t)numvec);
    k.GetConfig().SetLanczosIter((std::size_t)m);
}
static int pseudo_main()
{
    const std::size_t nev = 16, nex = 6, nevex = nev + nex;
    const int numvec = 4, m = 22;
    ScalarTape tape;
    SolveStats sa, sb, sc;
    Script a(2000, nev, nex, true), b(2000, nev, nex, true), c(2000, nev, nex, true);
    script(a, 911, nevex, numvec, m); script(b, 4242, nevex, numvec, m); script(c, 4242, nevex, numvec, m);
    {
        TapeKernel<double> t(&a, nullptr, &tape, TapeKernel<double>::RECORD);
        Algorithm<double, ChaseBase<double>>::solve_pseudo(&t, &sa);
    }
    std::printf("tape_size %zu\n", tape.data.size());
    std::printf("A iterations %zu filtered %zu locked %zu\n", sa.iterations, sa.filtered_vecs, sa.locked);
    for (const auto& l : a.calls) std::printf("A %s\n", l.c_str());
    {
        TapeKernel<double> t(&b, nullptr, &tape, TapeKernel<double>::REPLAY);
        Algorithm<double, ChaseBase<double>>::solve_pseudo(&t, &sb);
    }
    std::printf("B iterations %zu filtered %zu locked %zu tape_pos %zu\n", sb.iterations, sb.filtered_vecs, sb.locked, tape.pos);
    for (const auto& l : b.calls) std::printf("B %s\n", l.c_str());
    for (std::size_t i = 0; i < nev; ++i) std::printf("ritz %.17g %.17g\n", a.GetRitzv()[i], b.GetRitzv()[i]);
    Algorithm<double, ChaseBase<double>>::solve_pseudo(&c, &sc);
    std::printf("C iterations %zu filtered %zu locked %zu\n", sc.iterations, sc.filtered_vecs, sc.locked);
    for (const auto& l : c.calls) std::printf("C %s\n", l.c_str());
    ScalarTape cut;
    cut.data.assign(tape.data.begin(), tape.data.begin() + tape.data.size() / 2);
    Script d(2000, nev, nex, true);
    script(d, 4242, nevex, numvec, m);
    try {
        TapeKernel<double> t(&d, nullptr, &cut, TapeKernel<double>::REPLAY);
        Algorithm<double, ChaseBase<double>>::solve_pseudo(&t, nullptr);
        std::printf("truncated no-error\n");
    } catch (const std::exception& e) {
        std::printf("truncated %s\n", e.what());
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::strcmp(argv[1], "pseudo") == 0) return pseudo_main();
    const size_t N = argc > 1 ? std::atoi(argv[1]) : 96, nev = argc > 2 ? std::atoi(argv[2]) : 8,
                 nex = argc > 3 ? std::atoi(argv[3]) : 6;
    const int deg = argc > 4 ? std::atoi(argv[4]) : 10;
    ScalarTape tape;
    SolveStats sa, sb, sc;
    Mock a(N, nev, nex, clement_matrix(N, 0.0));
    a.GetConfig().SetDeg(deg);
    {
        TapeKernel<double> t(&a, nullptr, &tape, TapeKernel<double>::RECORD);
        Algorithm<double, ChaseBase<double>>::solve(&t, &sa);
    }
    std::printf("tape_size %zu\n", tape.data.size());
    std::printf("A iterations %zu filtered %zu locked %zu\n", sa.iterations, sa.filtered_vecs, sa.locked);
    for (const auto& l : a.calls) std::printf("A %s\n", l.c_str());

    Mock b(N, nev, nex, other_matrix(N));
    b.GetConfig().SetDeg(deg);
    {
        TapeKernel<double> t(&b, nullptr, &tape, TapeKernel<double>::REPLAY);
        Algorithm<double, ChaseBase<double>>::solve(&t, &sb);
    }
    std::printf("B iterations %zu filtered %zu locked %zu tape_pos %zu\n", sb.iterations, sb.filtered_vecs, sb.locked, tape.pos);
    for (const auto& l : b.calls) std::printf("B %s\n", l.c_str());
    for (size_t i = 0; i < nev; ++i) std::printf("ritz %.17g %.17g\n", a.GetRitzv()[i], b.GetRitzv()[i]);

    Mock c(N, nev, nex, other_matrix(N));
    c.GetConfig().SetDeg(deg);
    Algorithm<double, ChaseBase<double>>::solve(&c, &sc);
    std::printf("C iterations %zu filtered %zu locked %zu\n", sc.iterations, sc.filtered_vecs, sc.locked);
    for (const auto& l : c.calls) std::printf("C %s\n", l.c_str());

    ScalarTape cut;
    cut.data.assign(tape.data.begin(), tape.data.begin() + tape.data.size() / 2);
    Mock d(N, nev, nex, other_matrix(N));
    d.GetConfig().SetDeg(deg);
    try {
        TapeKernel<double> t(&d, nullptr, &cut, TapeKernel<double>::REPLAY);
        Algorithm<double, ChaseBase<double>>::solve(&t, nullptr);
        std::printf("truncated no-error\n");
    } catch (const std::exception& e) {
        std::printf("truncated %s\n", e.what());
    }
    return 0;
}
