// micro-benchmark: a chain of N small dependent kernels (the shape of the tridiagonalisation's per-column launches) issued
// one by one on a stream vs replayed from a captured hipGraph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void step(double* x, int n, int k) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = x[i] * 1.0000001 + k;
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 7680, blocks = argc > 2 ? atoi(argv[2]) : 20;
    double* x; hipMalloc(&x, 1 << 20); hipMemset(x, 0, 1 << 20);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    auto run = [&]() { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step, dim3(blocks), dim3(256), 0, st, x, 4096, k); };
    run(); hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    run(); hipStreamSynchronize(st);
    auto t1 = std::chrono::steady_clock::now();
    hipGraph_t g; hipGraphExec_t ge;
    auto c0 = std::chrono::steady_clock::now();
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    run();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    auto c1 = std::chrono::steady_clock::now();
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    auto t2 = std::chrono::steady_clock::now();
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    auto t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    printf("%d dependent launches of %d blocks: stream %.2f ms (%.2f us each); graph capture+instantiate %.2f ms, first replay %.2f ms, replay %.2f ms (%.2f us each)\n",
           N, blocks, ms(t0, t1), ms(t0, t1) * 1e3 / N, ms(c0, c1), ms(c1, t2), ms(t2, t3), ms(t2, t3) * 1e3 / N);
    return 0;
}
