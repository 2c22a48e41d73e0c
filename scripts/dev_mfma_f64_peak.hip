// Attainable rate of v_mfma_f64_16x16x4_f64 on gfx950: bare register-resident loops (no memory traffic), then the same
// loop with the LDS fragment reads / barriers of the GEMM kernel's inner loop added one by one.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/dev_mfma_f64_peak.hip && /tmp/mfma_peak
// Prints TFLOP/s over all CUs and shader cycles per MFMA per SIMD (s_memtime around the loop, median over waves) and the
// in-kernel clock (s_memtime / s_memrealtime x 100 MHz).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

// MODE 0: bare MFMAs on NACC independent accumulators
// MODE 1: + 6 ds_read_b128 per 24 MFMAs (operands come from LDS)
// MODE 2: MODE 1 + one s_barrier per 48 MFMAs
// PRIO 1: the workgroup whose TG_ID (HW_ID bits 19:16) is odd raises its priority; PRIO 2: the same on WAVE_ID (bits 3:0)
template <int NACC, int MODE, int PRIO = 0>
__global__ __launch_bounds__(256, 2) void mfma_loop(double* out, long long* stamps, int iters, double seed)
{
    __shared__ d2 lds[2048];
    const int tid = threadIdx.x;
    if constexpr (PRIO != 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned sel = PRIO == 1 ? (hw >> 16) & 1u : hw & 1u;
        if (sel) __builtin_amdgcn_s_setprio(1);
    }
    for (int i = tid; i < 2048; i += 256) lds[i] = d2{seed * (i + 1), seed * (i + 3)};
    __syncthreads();
    d4 acc[NACC];
    #pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a[6], b[6];
    #pragma unroll
    for (int i = 0; i < 6; ++i) { a[i] = seed * (tid + i); b[i] = seed * (tid - i); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE >= 1) {
            #pragma unroll
            for (int i = 0; i < 3; ++i) {
                const d2 va = lds[(tid + 64 * i + it * 7) & 2047];
                const d2 vb = lds[(tid * 3 + 64 * i + it * 5) & 2047];
                a[2 * i] = va.x; a[2 * i + 1] = va.y; b[2 * i] = vb.x; b[2 * i + 1] = vb.y;
            }
        }
        #pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i % 6], b[(i / 6) % 6], acc[i], 0, 0, 0);
        if constexpr (MODE >= 2) { if (it & 1) __syncthreads(); }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    d4 s = d4{0, 0, 0, 0};
    #pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[(size_t)blockIdx.x * 256 + tid] = s.x + s.y + s.z + s.w;
    if ((tid & 63) == 0) {
        stamps[2 * ((size_t)blockIdx.x * 4 + (tid >> 6))] = t1 - t0;
        stamps[2 * ((size_t)blockIdx.x * 4 + (tid >> 6)) + 1] = r1 - r0;
    }
}

__global__ __launch_bounds__(256, 2) void hwid_probe(unsigned* ids, int spin)
{
    __shared__ double pad[9000];                                   // 72 KB: two workgroups per CU like the GEMM kernel
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double x = pad[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;       // stay resident long enough for the whole grid to be placed
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) ids[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
    if (x == 12345.678) ids[0] = 0;
}

static void probe_hwid(int ncu)
{
    const int blocks = 2 * ncu;
    unsigned* d;
    CHECK(hipMalloc(&d, blocks * 4 * 4));
    hwid_probe<<<blocks, 256>>>(d, 200000);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    int tg_hist[16] = {0}, wave_hist[16] = {0}, same_tg_in_wg = 0, xcd_rr = 0;
    for (int b = 0; b < blocks; ++b) {
        bool same = true;
        for (int w = 0; w < 4; ++w) {
            const unsigned v = h[b * 4 + w];
            tg_hist[(v >> 16) & 15]++; wave_hist[v & 15]++;
            if (((v >> 16) & 15) != ((h[b * 4] >> 16) & 15)) same = false;
        }
        same_tg_in_wg += same;
    }
    std::printf("HW_ID probe (%d workgroups of 4 waves, 2 per CU): TG_ID histogram:", blocks);
    for (int i = 0; i < 16; ++i) if (tg_hist[i]) std::printf(" [%d]=%d", i, tg_hist[i]);
    std::printf("; WAVE_ID histogram:");
    for (int i = 0; i < 16; ++i) if (wave_hist[i]) std::printf(" [%d]=%d", i, wave_hist[i]);
    std::printf("; workgroups whose 4 waves share one TG_ID: %d\n first 12 workgroups (hex HW_ID of wave 0..3):", same_tg_in_wg);
    for (int b = 0; b < 12; ++b) std::printf(" | %08x %08x %08x %08x", h[b * 4], h[b * 4 + 1], h[b * 4 + 2], h[b * 4 + 3]);
    std::printf("\n");
    (void)xcd_rr;
    CHECK(hipFree(d));
}

template <int NACC, int MODE, int PRIO = 0>
static void run(const char* name, int wgs_per_cu, int ncu)
{
    const int iters = 20000;
    const int blocks = ncu * wgs_per_cu;
    double* out; long long* st;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    CHECK(hipMalloc(&st, (size_t)blocks * 4 * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) mfma_loop<NACC, MODE, PRIO><<<blocks, 256>>>(out, st, iters, 1e-3);
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) mfma_loop<NACC, MODE, PRIO><<<blocks, 256>>>(out, st, iters, 1e-3);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h((size_t)blocks * 8);
    CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (size_t i = 0; i < h.size() / 2; ++i) {
        cyc.push_back((double)h[2 * i]);
        clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);     // GHz: realtime ticks at 100 MHz
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double mfmas_per_wave = (double)iters * NACC;
    const double flops = (double)reps * blocks * 4 * mfmas_per_wave * 2048.0;
    const double cyc_per_mfma_simd = cyc[cyc.size() / 2] / (mfmas_per_wave * wgs_per_cu);   // wgs_per_cu waves share a SIMD
    std::printf("%-44s waves/SIMD %d  %8.2f TFLOP/s  %6.2f cyc/MFMA/SIMD  clock %.3f GHz\n", name, wgs_per_cu,
                flops / (ms * 1e-3) / 1e12, cyc_per_mfma_simd, clk[clk.size() / 2]);
    CHECK(hipFree(out)); CHECK(hipFree(st));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    std::printf("%s, %d CUs, clock %d MHz: nominal fp64 MFMA peak %.1f TFLOP/s (128 flop/clk/CU)\n", p.name, ncu,
                p.clockRate / 1000, ncu * 128.0 * p.clockRate * 1e3 / 1e12);
    for (int w = 1; w <= 2; ++w) {
        run<24, 0>("bare, 24 independent accumulators", w, ncu);
        run<12, 0>("bare, 12 independent accumulators", w, ncu);
        run<4, 0>("bare, 4 independent accumulators", w, ncu);
        run<1, 0>("bare, 1 accumulator (dependent chain)", w, ncu);
        run<24, 1>("24 acc + 6 ds_read_b128 per 24 MFMAs", w, ncu);
        run<24, 2>("24 acc + LDS reads + barrier per 48 MFMAs", w, ncu);
    }
    probe_hwid(ncu);
    run<24, 1, 1>("LDS reads, priority by TG_ID parity", 2, ncu);
    run<24, 2, 1>("LDS reads + barrier, priority by TG_ID parity", 2, ncu);
    run<24, 1, 2>("LDS reads, priority by WAVE_ID parity", 2, ncu);
    run<24, 2, 2>("LDS reads + barrier, priority by WAVE_ID parity", 2, ncu);
    return 0;
}
