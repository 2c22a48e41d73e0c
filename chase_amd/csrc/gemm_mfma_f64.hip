// gemm_mfma_f64.hip — fp64 / complex-fp64 panel GEMM for gfx950 (MI355X, CDNA4)
//
//   C[m x n] = alpha * op(A) * B + beta * C        column-major, op in {N, C}
//
// This is the kernel behind every dense product of the ChASE hot path:
//   * Chebyshev-filter HEMM      V2 = a*(H-cI)*V1 + b*V2   (reference: Impl/chase_cpu/chase_cpu.hpp:449-508,
//                                                            Impl/chase_gpu/chase_gpu.hpp:656-678 -> cublasTgemm N,N)
//   * distributed HEMM           W = a*H_loc^H*V + b*W      (reference: linalg/internal/mpi/hemm.hpp:114-229 -> gemm C,N / N,N)
//   * Gram / projection products A = V^H*V, A = W^H*Q       (reference: linalg/internal/cpu/cholqr1.hpp:41 (herk),
//                                                            linalg/internal/cpu/rayleighRitz.hpp:84-88)
//   * back-transform             V = Q*A                    (reference: linalg/internal/cpu/rayleighRitz.hpp:110)
//
// Design (MI355X-first, not a translation of any CUDA tiling):
//   * v_mfma_f64_16x16x4_f64, one 64-lane wave owns a (16*TM) x (16*TN) output tile; the 4 waves of a workgroup are
//     stacked along M (4x1) and each spans the tile's whole width, so skipping the 16-column groups of a ragged last
//     column tile unloads all four SIMDs alike.
//     MFMA operand roles are swapped (first operand = B/V fragment, second = A/H fragment) so that the accumulator's
//     lane index runs along the column-major M direction -> 256-byte contiguous C stores per 16 lanes.
//   * Everything in LDS is addressed in 16-byte "units" so that every fragment fetch is one ds_read_b128 and the
//     16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots (conflict-free without padding):
//       - real,    M-contiguous operand (A, op=N): unit = rows (2r, 2r+1) at one k  -> feeds two M tiles at once
//       - real,    K-contiguous operand (B; A op=C): unit = k (2q, 2q+1) of one row -> feeds two consecutive MFMAs
//       - complex: unit = one (re, im) element
//     The K order inside an 8-deep (real) chunk is permuted (MFMA s takes k = 2q+s) identically for both operands.
//   * K-contiguous tiles are stored [k-unit][row ^ k-unit] (XOR on the low 3 bits) so that the staging ds_write_b128
//     and the fragment ds_read_b128 are both conflict-free.
//   * complex: 4 real MFMAs per (re,im) tile step on planar fragments held in registers (interleaved in HBM/LDS); filter
//     products: 3 real MFMAs (the "3M" scheme, see the kernel's M3 parameter).
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4 in its scalar-base + 32-bit lane-offset form, inline assembly: no
//     64-bit vector address arithmetic, which on gfx950 runs on the fp64 matrix unit), 3 (complex) / 2 (real) LDS stages, the
//     copies of the next tile hung one by one under the MFMA groups of the current one, software-pipelined fragment reads,
//     one barrier per K step, two workgroups per CU; ragged M / unaligned operands / partial K tiles fall back to a
//     register-staged path (global -> register -> LDS) in the same kernel.  profiles/r02_mfma_f64_issue.txt has the numbers.
//   * XCD-aware tile order: consecutive logical tiles (same A row panel, different column panels) run on one XCD
//     so the streamed H panel is fetched once per XCD L2.
//   * deterministic split-K (slabs + fixed-order reduce) for the short-and-fat Gram products (k = N >> m, n).
//
// No vendor BLAS, no CUDA headers.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "kernels.h"

#ifndef CHASE_M3_PIPELINE
#define CHASE_M3_PIPELINE 1
#endif
// round 6: the V-side operand sums of the 3M kernels (br + bi) come from a PRECOMPUTED plane instead of four v_add_f64 per MFMA
// cluster (on gfx950 fp64 vector adds execute on the matrix unit: ~8 cycles of the pipe each, profiles/r02_mfma_f64_issue.txt).
// 0 builds round 2-5's loop for comparison (scripts/dev_build_variant.sh).
#ifndef CHASE_M3_SPLANE
#define CHASE_M3_SPLANE 1
#endif
// K steps a tile of the plane-fed loop is requested ahead of its first use: 2 (three LDS stages in flight); 1 = experiment of round 6
// (tiles one step ahead like the plane: neither faster nor slower on any device tried, profiles/r06_tile_group.txt)
#ifndef CHASE_M3_DEPTH
#define CHASE_M3_DEPTH 2
#endif

namespace chase_hip {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef d2_t d2u_t __attribute__((aligned(8)));   // 16-byte vector that may sit on an 8-byte boundary in HBM

// NARROW (real only): a 128 x 64 block tile instead of 128 x 128.  Its stage is 24 KB like the complex kernels', so three
// stages fit twice into a CU's 160 KB of LDS and the K loop is software-pipelined the same way (two tiles in flight, fragments
// double-buffered next to 64 accumulator registers).  It serves the widths the 128-wide tile handles badly: blocks narrower
// than a tile and the ragged rest of a width - launches that are bound by how fast H streams in, not by the matrix cores.
template <bool CPLX, bool OPA_C, bool NARROW = false>
struct Cfg {
    static_assert(!(CPLX && NARROW), "the complex tile is 64 columns wide already");
    static constexpr int NTHREADS = 256;
    // the 4 waves are stacked along M and every wave spans the tile's whole width, so the 16-column groups a
    // ragged last column tile does not need are skipped by all four waves alike (balanced over the 4 SIMDs)
    static constexpr int WAVES_M = 4, WAVES_N = 1;
    static constexpr int TM = 2;                    // 16-row MFMA tiles per wave along M
    static constexpr int TN = (CPLX || NARROW) ? 4 : 8;   // 16-col MFMA tiles per wave along N
    static constexpr int BM = 16 * TM * WAVES_M;    // 128
    static constexpr int BN = 16 * TN * WAVES_N;    // 128 (real) / 64 (complex, narrow real)
    static constexpr int BK = CPLX ? 8 : 16;        // elements of T along K per stage
    static constexpr int KPU = CPLX ? 1 : 2;        // k per 16-byte unit of a K-contiguous operand
    static constexpr int RPU = CPLX ? 1 : 2;        // rows per 16-byte unit of an M-contiguous operand
    static constexpr int A_UNITS = OPA_C ? BM * 8 : BK * (BM / RPU);   // 1024 in all variants
    static constexpr int B_UNITS = BN * 8;                              // 1024 real / 512 complex
    static constexpr int A_PASSES = A_UNITS / NTHREADS;
    static constexpr int B_PASSES = B_UNITS / NTHREADS;
    static constexpr int STAGE_UNITS = A_UNITS + B_UNITS;
    static constexpr int EPT = CPLX ? 2 : 1;        // doubles per element
    // complex: 3 LDS stages filled by global_load_lds two K steps ahead (72 KB per workgroup, two workgroups per CU);
    // real: 2 stages through registers (3 x 32 KB x 2 workgroups would not fit the 160 KB LDS)
    static constexpr int STAGES = (CPLX || NARROW) ? 3 : 2;
    static constexpr bool PIPELINED = CPLX || NARROW;   // fragments double-buffered, barrier between the two MFMA clusters
    static constexpr int A_GLDS = A_UNITS / 64, B_GLDS = B_UNITS / 64;     // wave instructions per tile
    static constexpr int GLDS_PER_WAVE = (A_GLDS + B_GLDS) / 4;
};

// LDS image of a K-contiguous operand tile (rows x 8 sixteen-byte units): unit ku of row r sits at r*8 + (ku ^ f(r)),
// f(r) = (r >> 1) & 7.  One wave instruction of 64 x 16 B covers 8 whole rows, so the image can be filled either by
// ds_write_b128 (8-lane groups write one contiguous 128-B row) or directly by global_load_lds (lane l <- row l/8, unit
// (l%8) ^ f(row): the swizzle lives on the SOURCE address); the 16-lane groups of the fragment ds_read_b128 hit 16
// distinct 16-byte slots (conflict-free, SQ_LDS_BANK_CONFLICT = 0 measured).
__device__ __forceinline__ int kidx(int r, int ku) { return r * 8 + (ku ^ ((r >> 1) & 7)); }

// bijective XCD remap (blocks b and b+8 share an XCD): gives each XCD a contiguous range of logical tiles
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblk)
{
    const unsigned q = nblk >> 3, r = nblk & 7u, x = b & 7u;
    const unsigned base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (b >> 3);
}

// Logical tile index -> (row panel bm, column tile bn).  Tiles are ordered in groups of GR row panels and, inside a
// group, column tile by column tile (GR = 1: row-major order of tiles).  A run of 64 consecutive tiles - what the 32 CUs x 2
// workgroups of one XCD hold at a time (xcd_remap gives every XCD a contiguous range) - then covers GR row panels x 64/GR
// column tiles.  Measured at N = 65536, n = 2560 (profiles/r02_tile_order.txt): GR = 2 and 4 are 0.9 % faster than GR = 1,
// GR = 8 is no faster and moves MORE bytes through the fabric (L2 misses: workgroups that share an H panel stay in step
// only while they were dispatched together) - the launch is bound by MFMA issue, not by the 1.7-2.5 TB/s of L2 fills.
// Default GR = 2, 4 for long launches with many column tiles (launch_gemm_part; CHASE_HIP_TILE_GROUP overrides).
__host__ __device__ __forceinline__ void tile_coords(int t, int gm, int gn, int GR, int& bm, int& bn)
{
    const int gsz = GR * gn;
    const int g = t / gsz;
    const int first = g * GR;
    const int rows = (gm - first < GR) ? gm - first : GR;
    const int r = t - g * gsz;
    bm = first + r % rows;
    bn = r / rows;
}

struct GemmArgs {
    const double* A; const double* B; double* C;     // C may point to split-K slabs
    long lda, ldb, ldc;                              // in elements of T
    int m, n, k;
    int gm, gn;                                      // output tiles
    int group_rows;                                  // tile order: row panels per group (1 = row-major order of tiles)
    int bn_cols;                                     // column stride of the output tiles: BN, or 16 g < BN ("uniform ragged" tiling)
    // Work decomposition (removes wave quantisation for arbitrary active widths): blocks [0, full_tiles) own one whole
    // output tile each; the remaining tiles ("tail": fewer than one full round of the chip) are cut into tail_sk K pieces
    // of tail_kchunk each, written as raw BM x BN partial slabs and combined by tail_reduce_kernel in a fixed order.
    int full_tiles, tail_sk, tail_kchunk;
    double* slabs;
    int glds_ok;                                     // operands are 16-byte addressable: direct global -> LDS copies allowed
    // 3M kernels: plane of the V-side operand sums br + bi in the kernel's own LDS image - per column tile bn and K step kt one
    // 4 KB block [4 k pairs][64 columns][2] of doubles at S + (bn * s_nkt + kt) * 512 (splane_kernel writes it before the launch)
    const double* S; int s_nkt;
    double alpha_re, alpha_im, beta_re, beta_im;
};

// TAG only changes the symbol name: TAG = 1 is the instantiation launched between FilterPhaseStart/End, so that
// rocprofv3 --kernel-trace --stats reports the Chebyshev-filter HEMM separately from the QR / RR / residual products.
// RAGGED: the launch covers a column range whose (single) last column tile needs fewer than TN 16-column groups; only
// that instantiation carries the per-group branches (the whole-tile one keeps its MFMA clusters branch-free).
// M3 (complex only): three real products per complex product instead of four (the "3M" scheme of zgemm3m),
//   P1 = sum ar*br, P2 = sum ai*bi, P3 = sum (ar +- ai)(br + bi);  op=N: re = P1 - P2, im = P3 - P1 - P2;
//   op=C (conj(A)): re = P1 + P2, im = P3 - P1 + P2  with (ar - ai).
// 25 % fewer MFMAs for the same product; normwise backward stable like the 4-product form (the imaginary part carries the
// absolute error of the real part), used for the Chebyshev-filter products only.
template <bool CPLX, bool OPA_C, int TAG, bool RAGGED, bool M3, bool NARROW = false>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs p)
{
    static_assert(CPLX || !M3, "3M applies to complex products");
    using C_ = Cfg<CPLX, OPA_C, NARROW>;
    constexpr int BM = C_::BM, BN = C_::BN, BK = C_::BK, TM = C_::TM, TN = C_::TN;
    constexpr int EPT = C_::EPT, KPU = C_::KPU, RPU = C_::RPU;
    constexpr int UM = BM / RPU;                    // units per k-column of an M-contiguous A tile

    extern __shared__ __attribute__((aligned(16))) d2_t lds[];   // [STAGES][STAGE_UNITS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % C_::WAVES_M, wn = wave / C_::WAVES_M;     // wave position in the WAVES_M x WAVES_N grid
    const int c16 = lane & 15, q = lane >> 4;

    // ---- logical tile -------------------------------------------------------------------------------------------
    unsigned tile;
    int kbeg = 0, kend = p.k;
    bool raw = false;                               // true: this block produces a partial slab of a tail tile
    double* slab = nullptr;
    if (blockIdx.x < (unsigned)p.full_tiles) {
        tile = xcd_remap(blockIdx.x, (unsigned)p.full_tiles);
    } else {
        const unsigned P = xcd_remap(blockIdx.x - (unsigned)p.full_tiles, gridDim.x - (unsigned)p.full_tiles);
        tile = (unsigned)p.full_tiles + P / (unsigned)p.tail_sk;
        const int bz = (int)(P % (unsigned)p.tail_sk);
        kbeg = bz * p.tail_kchunk;
        kend = min(p.k, kbeg + p.tail_kchunk);
        raw = true;
        slab = p.slabs + (size_t)P * (BM * BN * EPT);
    }
    int bm, bn;
    tile_coords((int)tile, p.gm, p.gn, p.group_rows, bm, bn);
    const int row0 = bm * BM, col0 = bn * p.bn_cols;
    const int nkt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

    const bool interior = (row0 + BM <= p.m) && (col0 + BN <= p.n);

    // ---- staging registers --------------------------------------------------------------------------------------
    d2_t ra[C_::A_PASSES], rb[C_::B_PASSES];

    auto load_tile = [&](int k0) {
        const bool kfull = (k0 + BK <= kend);
        if (interior && kfull) {
            #pragma unroll
            for (int ps = 0; ps < C_::A_PASSES; ++ps) {
                const int idx = ps * 256 + tid;
                if constexpr (!OPA_C) {
                    const int kk = idx / UM, u = idx % UM;
                    const double* g = p.A + ((long)(k0 + kk) * p.lda + row0 + u * RPU) * EPT;
                    ra[ps] = *(const d2u_t*)g;
                } else {
                    const int r = idx >> 3, ku = idx & 7;
                    const double* g = p.A + ((long)(row0 + r) * p.lda + k0 + ku * KPU) * EPT;
                    ra[ps] = *(const d2u_t*)g;
                }
            }
            #pragma unroll
            for (int ps = 0; ps < C_::B_PASSES; ++ps) {
                const int idx = ps * 256 + tid;
                const int r = idx >> 3, ku = idx & 7;
                const double* g = p.B + ((long)(col0 + r) * p.ldb + k0 + ku * KPU) * EPT;
                rb[ps] = *(const d2u_t*)g;
            }
        } else {
            #pragma unroll
            for (int ps = 0; ps < C_::A_PASSES; ++ps) {
                const int idx = ps * 256 + tid;
                d2_t v = {0.0, 0.0};
                if constexpr (!OPA_C) {
                    const int kk = idx / UM, u = idx % UM;
                    const int gi = row0 + u * RPU, gk = k0 + kk;
                    if (gk < kend) {
                        const double* g = p.A + ((long)gk * p.lda + gi) * EPT;
                        if constexpr (CPLX) { if (gi < p.m) v = *(const d2u_t*)g; }
                        else { if (gi < p.m) v.x = g[0]; if (gi + 1 < p.m) v.y = g[1]; }
                    }
                } else {
                    const int r = idx >> 3, ku = idx & 7;
                    const int gi = row0 + r, gk = k0 + ku * KPU;
                    if (gi < p.m) {
                        const double* g = p.A + ((long)gi * p.lda + gk) * EPT;
                        if constexpr (CPLX) { if (gk < kend) v = *(const d2u_t*)g; }
                        else { if (gk < kend) v.x = g[0]; if (gk + 1 < kend) v.y = g[1]; }
                    }
                }
                ra[ps] = v;
            }
            #pragma unroll
            for (int ps = 0; ps < C_::B_PASSES; ++ps) {
                const int idx = ps * 256 + tid;
                const int r = idx >> 3, ku = idx & 7;
                const int gj = col0 + r, gk = k0 + ku * KPU;
                d2_t v = {0.0, 0.0};
                if (gj < p.n) {
                    const double* g = p.B + ((long)gj * p.ldb + gk) * EPT;
                    if constexpr (CPLX) { if (gk < kend) v = *(const d2u_t*)g; }
                    else { if (gk < kend) v.x = g[0]; if (gk + 1 < kend) v.y = g[1]; }
                }
                rb[ps] = v;
            }
        }
    };

    auto store_tile = [&](int stage) {
        d2_t* sA = lds + stage * C_::STAGE_UNITS;
        d2_t* sB = sA + C_::A_UNITS;
        #pragma unroll
        for (int ps = 0; ps < C_::A_PASSES; ++ps) {
            const int idx = ps * 256 + tid;
            if constexpr (!OPA_C) {
                sA[idx] = ra[ps];                                   // [k][unit], identical to the load order
            } else {
                const int r = idx >> 3, ku = idx & 7;
                sA[kidx(r, ku)] = ra[ps];
            }
        }
        #pragma unroll
        for (int ps = 0; ps < C_::B_PASSES; ++ps) {
            const int idx = ps * 256 + tid;
            const int r = idx >> 3, ku = idx & 7;
            sB[kidx(r, ku)] = rb[ps];
        }
    };

    // ---- accumulators -------------------------------------------------------------------------------------------
    constexpr int NACC = CPLX ? (M3 ? 3 : 2) : 1;
    d4_t acc[NACC][TN][TM];
    #pragma unroll
    for (int z = 0; z < NACC; ++z)
        #pragma unroll
        for (int j = 0; j < TN; ++j)
            #pragma unroll
            for (int i = 0; i < TM; ++i) acc[z][j][i] = d4_t{0.0, 0.0, 0.0, 0.0};

    const int wrow = wm * (16 * TM);                // wave's first row inside the block tile
    const int wcol = wn * (16 * TN);
    // ragged last column tile: 16-column groups past n carry no MFMAs (wave-uniform count, 0..TN), so a partial tile
    // costs what its valid columns cost and the freed matrix-core time goes to the co-resident workgroup
    const int jv = RAGGED ? __builtin_amdgcn_readfirstlane(min(p.bn_cols >> 4, max(0, (p.n - col0 - wcol + 15) >> 4))) : TN;

    // Fragments of one 4-unit K chunk: b[j] = 16-byte unit of column tile j, a[i][.] = the two doubles of row tile i's unit
    // (complex: re, im of one element; real: the chunk's two k values s = 0, 1).
    struct Frag { d2_t b[TN]; double a[TM][2]; };

    auto read_chunk = [&](int stage, int ch, Frag& f) __attribute__((always_inline)) {
        const d2_t* sA = lds + stage * C_::STAGE_UNITS;
        const d2_t* sB = sA + C_::A_UNITS;
        const int ku = 4 * ch + q;
        // A fragments first: LDS returns in order, so the first MFMA can start after three reads instead of five
        if constexpr (CPLX || OPA_C) {
            #pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wrow + 16 * i + c16;
                d2_t v;
                if constexpr (OPA_C) v = sA[kidx(r, ku)];
                else                 v = sA[ku * UM + r];
                f.a[i][0] = v.x; f.a[i][1] = v.y;
            }
            #pragma unroll
            for (int j = 0; j < TN; ++j) f.b[j] = sB[kidx(wcol + 16 * j + c16, ku)];
        } else {
            #pragma unroll
            for (int j = 0; j < TN; ++j) f.b[j] = sB[kidx(wcol + 16 * j + c16, ku)];
            // real, M-contiguous A: unit = rows (2u, 2u+1) at one k; MFMA step s uses k = 8*ch + 2*q + s
            #pragma unroll
            for (int pr = 0; pr < TM / 2; ++pr)
                #pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int kk = 8 * ch + 2 * q + s;
                    const d2_t v = sA[kk * UM + (wrow / 2) + 16 * pr + c16];
                    f.a[2 * pr][s] = v.x;
                    f.a[2 * pr + 1][s] = v.y;
                }
        }
    };

    // `hook(slot)` runs after every group of MFMAs (4M: 4 MFMAs, real: TM, 3M: 3; slots 0..2 TN-1 in each case): the
    // LDS-DMA loops hang the global -> LDS copies of the next tile there, one copy under each group's MFMAs
    auto no_hook = [](int) __attribute__((always_inline)) {};
    auto mfma_chunk = [&](const Frag& f, auto&& hook) __attribute__((always_inline)) {
        constexpr bool FULL = !RAGGED;                        // FULL: every 16-column group is live (no per-group branches)
        if constexpr (CPLX && M3) {
            double sa[TM], sb[TN];
            #pragma unroll
            for (int i = 0; i < TM; ++i) sa[i] = OPA_C ? f.a[i][0] - f.a[i][1] : f.a[i][0] + f.a[i][1];
            #pragma unroll
            for (int j = 0; j < TN; ++j) sb[j] = f.b[j].x + f.b[j].y;
            #pragma unroll
            for (int j = 0; j < TN; ++j) {
                #pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (FULL || j < jv) {
                        acc[0][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.b[j].x, f.a[i][0], acc[0][j][i], 0, 0, 0);
                        acc[1][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.b[j].y, f.a[i][1], acc[1][j][i], 0, 0, 0);
                        acc[2][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(sb[j], sa[i], acc[2][j][i], 0, 0, 0);
                    }
                    hook(TM * j + i);
                }
            }
        } else if constexpr (CPLX) {
            // op=N: (ar + i ai)(br + i bi): re = br ar - bi ai, im = bi ar + br ai
            // op=C: (ar - i ai)(br + i bi): re = br ar + bi ai, im = bi ar - br ai      -> one negated B value per tile
            double nb[TN];
            #pragma unroll
            for (int j = 0; j < TN; ++j) nb[j] = OPA_C ? -f.b[j].x : -f.b[j].y;
            #pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (FULL || j < jv)
                #pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[0][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.b[j].x, f.a[i][0], acc[0][j][i], 0, 0, 0);
                    acc[1][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.b[j].y, f.a[i][0], acc[1][j][i], 0, 0, 0);
                }
                hook(j);
            }
            #pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (FULL || j < jv)
                #pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[0][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(OPA_C ? f.b[j].y : nb[j], f.a[i][1], acc[0][j][i], 0, 0, 0);
                    acc[1][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(OPA_C ? nb[j] : f.b[j].x, f.a[i][1], acc[1][j][i], 0, 0, 0);
                }
                hook(TN + j);
            }
        } else {
            #pragma unroll
            for (int s = 0; s < 2; ++s)
                #pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (FULL || j < jv)
                    #pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[0][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            s == 0 ? f.b[j].x : f.b[j].y, f.a[i][s], acc[0][j][i], 0, 0, 0);
                    hook(s * TN + j);
                }
        }
    };

    // whole K step from one LDS stage (register-staged fallback path)
    auto compute = [&](int stage, auto&& hook) __attribute__((always_inline)) {
        if constexpr (C_::PIPELINED && !M3) {
            Frag f0, f1;
            read_chunk(stage, 0, f0);
            read_chunk(stage, 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk(f0, hook);
            mfma_chunk(f1, no_hook);
        } else {                                             // 40 fragment registers per chunk: one chunk at a time
            {
                Frag f;
                read_chunk(stage, 0, f);
                mfma_chunk(f, hook);
            }
            {
                Frag f;
                read_chunk(stage, 1, f);
                mfma_chunk(f, no_hook);
            }
        }
    };

    // ---- main loop ----------------------------------------------------------------------------------------------
    bool done = false;
    {
        // Interior workgroups: asynchronous global -> LDS copies (global_load_lds_dwordx4, no staging registers, no
        // ds_write pass); complex: three LDS stages, tile kt+2 in flight while tile kt is multiplied; real: two stages,
        // tile kt+1 in flight.  One raw s_barrier and one COUNTED vmcnt per K step.
        const int nfull = (kend - kbeg) / BK;
        // column-edge workgroups qualify too: B rows past n are clamped to the last valid column (their products are
        // never stored), only a ragged M edge needs the guarded register path
        if ((row0 + BM <= p.m) && nfull >= 1 && p.glds_ok) {
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            // Source addresses = wave-uniform 64-bit base (SGPRs, advanced by a constant stride per K step with two scalar
            // adds) + a per-lane 32-bit byte offset that never changes: no 64-bit vector arithmetic or multiplies in the
            // loop.  The LDS stage cycles through a running counter.
            constexpr int NA = C_::A_GLDS / 4, NB = C_::B_GLDS / 4;
            const char* sa[NA];                                 // uniform
            unsigned va[NA];                                    // per lane, bytes
            // diagnostic builds (timing only, results wrong on purpose; scripts/dev_build_variant.sh): every workgroup streams
            // the SAME A row panel / the same B column panel, so that operand is served by the L2s - what would a launch gain if
            // its re-reads never left the L2?  (profiles/r04_traffic_upper_bound.txt)
#ifdef CHASE_DIAG_SAME_A
            const int row0_src = 0;
#else
            const int row0_src = row0;
#endif
#ifdef CHASE_DIAG_SAME_B
            const int col0_src = 0;
#else
            const int col0_src = col0;
#endif
            const char* sb = (const char*)(p.B + ((long)col0_src * p.ldb + kbeg) * EPT);
            unsigned vb[NB];
            #pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int t = wv * NA + u;
                if constexpr (!OPA_C) {
                    // [k][unit] image: complex 128 units per k row (two instructions), real 64 units (one)
                    const int kk = CPLX ? (t >> 1) : t, half = CPLX ? (t & 1) : 0;
                    sa[u] = (const char*)(p.A + ((long)(kbeg + kk) * p.lda + row0_src) * EPT + (long)half * 128);
                    va[u] = (unsigned)lane * 16u;
                } else {
                    // row r = 8 t + lane / 8: the 8 t rows go into the uniform base; the swizzle (r >> 1) & 7 = (4 t + lane / 16) & 7
                    // depends on t through its parity only, so the lane offsets of copies u and u + 2 are the same register
                    const int rl = lane >> 3, ku = (lane & 7) ^ ((4 * t + (lane >> 4)) & 7);
                    sa[u] = (const char*)(p.A + ((long)(row0_src + 8 * t) * p.lda + kbeg) * EPT);
                    va[u] = (unsigned)(((long)rl * p.lda + ku * KPU) * EPT * 8);
                }
            }
            #pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int t = wv * NB + u;
                const int r = 8 * t + (lane >> 3), ku = (lane & 7) ^ ((r >> 1) & 7);
                const int rc = min(r, p.n - 1 - col0);          // rows past n re-read the last valid column
                vb[u] = (unsigned)(((long)rc * p.ldb + ku * KPU) * EPT * 8);
            }
            const long stepA = (OPA_C ? (long)BK * EPT : (long)BK * p.lda * EPT) * 8;
            constexpr long stepB = (long)BK * EPT * 8;
            int st_issue = 0;                                   // stage the next issue() fills
            // One global -> LDS copy instruction (64 lanes x 16 B) of the tile being requested.  Written as inline assembly
            // for the SGPR-base + 32-bit-lane-offset form of global_load_lds_dwordx4: the compiler's intrinsic forms a
            // 64-bit vector address first (v_lshl_add_u64), and on gfx950 64-bit vector integer adds - like v_add_f64 - run on
            // the unit that executes v_mfma_f64, so every such add is taken straight out of the matrix pipe's time
            // (profiles/r02_mfma_f64_issue.txt).  M0 = LDS destination of lane 0.  Every LDS-DMA of this kernel goes through
            // here, so the compiler never holds a value of its own in M0.
            auto issue_one = [&](int u, int stage) __attribute__((always_inline)) {
                // (the plane-fed 3M kernels keep all A stages together, then all B stages: one LDS base register per fragment
                // pattern reaches every stage through the 16-bit immediate offset of ds_read)
                constexpr bool SPLIT = CPLX && M3 && CHASE_M3_SPLANE;
                d2_t* sA = SPLIT ? lds + stage * C_::A_UNITS : lds + stage * C_::STAGE_UNITS;
                d2_t* sB = SPLIT ? lds + C_::STAGES * C_::A_UNITS + stage * C_::B_UNITS : sA + C_::A_UNITS;
                if (u < NA) {
                    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(sA + (wv * NA + u) * 64);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(va[u]), "s"(sa[u]), "s"(dst) : "memory");
                    sa[u] += stepA;
                } else {
                    const int ub = u - NA;
                    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(sB + (wv * NB + ub) * 64);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(vb[ub]), "s"(sb), "s"(dst) : "memory");
                }
                if (u == NA + NB - 1) sb += stepB;
            };
            auto issue = [&]() __attribute__((always_inline)) {
                #pragma unroll
                for (int u = 0; u < NA + NB; ++u) issue_one(u, st_issue);
                st_issue = (st_issue + 1 == C_::STAGES) ? 0 : st_issue + 1;
            };
            if constexpr (C_::PIPELINED && !M3) {
                // Software pipeline (complex four-product kernel and the narrow real kernel): all STAGES tiles are requested up front; each K step multiplies chunk 0 from registers
                // while chunk 1's fragments stream in from LDS, and the barrier that publishes tile kt+1 sits BETWEEN the two
                // MFMA clusters, so chunk 0 of tile kt+1 is fetched under chunk 1's MFMAs and the stage of tile kt is refilled
                // (tile kt+STAGES) as soon as its last fragment has been read: no MFMA ever waits for an LDS read.
                constexpr int G = C_::GLDS_PER_WAVE;
                const int npre = min(nfull, C_::STAGES);
                for (int t = 0; t < npre; ++t) issue();
                if (npre == 3)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
                else if (npre == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
                else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                Frag fA, fB;
                read_chunk(0, 0, fA);
                constexpr std::true_type yes{};
                constexpr std::false_type no{};
                // one K step on the tile in stage st (not the last one); `fill`: tile kt+STAGES exists and goes into stage st,
                // one copy after every fourth MFMA of the second cluster
                auto kstep = [&](int kt, int st, auto fill_c) __attribute__((always_inline)) {
                    const int stn = (st + 1 == C_::STAGES) ? 0 : st + 1;
                    read_chunk(st, 1, fB);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_chunk(fA, no_hook);
                    __builtin_amdgcn_sched_barrier(0);
                    // my reads of stage st are complete (it is refilled below) and my copies of tile kt+1 have landed;
                    // a later tile may stay in flight (three stages)
                    if (C_::STAGES > 2 && (decltype(fill_c)::value || kt + 2 < nfull))
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    read_chunk(stn, 0, fA);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_chunk(fB, [&](int slot) __attribute__((always_inline)) {
                        if (decltype(fill_c)::value && slot < NA + NB) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_one(slot, st);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                };
                static_assert(2 * TN >= NA + NB, "one copy per MFMA group of the cluster");
                int st = 0, kt = 0;
                for (; kt + C_::STAGES < nfull; ++kt) { kstep(kt, st, yes); st = (st + 1 == C_::STAGES) ? 0 : st + 1; }
                for (; kt + 1 < nfull; ++kt)          { kstep(kt, st, no);  st = (st + 1 == C_::STAGES) ? 0 : st + 1; }
                read_chunk(st, 1, fB);                                         // last tile: nothing to publish or prefetch
                __builtin_amdgcn_sched_barrier(0);
                mfma_chunk(fA, no_hook);
                __builtin_amdgcn_sched_barrier(0);
                mfma_chunk(fB, no_hook);
            } else if constexpr (CPLX && M3 && CHASE_M3_SPLANE) {
                // 3M software pipeline, round 6: V-side operand sums from the precomputed plane p.S.
                //   * LDS: the three 24 KB stages + a RING OF TWO 4 KB stages for the plane = 80 KB (two workgroups still fit
                //     the CU's 160 KB).  S(kt) lives in ring stage kt & 1, image [4 k pairs][64 columns][2] doubles, brought in by ONE
                //     more global -> LDS copy per wave and K step (1 KB of the 4 KB block each, linear: the plane is stored in
                //     this image).  S(kt + 2) is requested right after the mid-step barrier of step kt (all reads of S(kt)
                //     have completed by then), BEFORE the copies of tile kt + 3, so that the counted vmcnt of the next step
                //     covers it: the waits are the ones of rounds 2-5.
                //   * MFMA order inside a cluster: row tile by row tile (i outer).  An A fragment is then REFILLED IN PLACE
                //     after its 12 MFMAs like the B fragments after theirs (no second A buffer: 8 registers, which the four
                //     prefetched sum fragments take); every refill is issued >= 9 MFMAs (576 cycles) before its first reader.
                //   * per cluster: 24 MFMAs, 2 v_add_f64 (the H-side sums ar +- ai, per wave private rows: they cannot be
                //     shared) instead of 6, 6 ds_read_b128 + 4 ds_read_b64.
                // Results are bitwise those of rounds 2-5: the plane holds the same IEEE sums br + bi, the accumulators are
                // independent of each other, so the MFMA order does not matter.
                constexpr int G = C_::GLDS_PER_WAVE;
                constexpr int S_UNITS = 256;                                   // 16-byte units per ring stage (4 KB)
                d2_t* const sring = lds + C_::STAGES * C_::STAGE_UNITS;
                const char* ssrc = (const char*)(p.S + ((size_t)bn * (size_t)p.s_nkt + (size_t)(kbeg / BK)) * 512) + wv * 1024;
                const unsigned vs = (unsigned)lane * 16u;
                int s_issue = 0;                                               // ring stage the next plane copy fills
                auto issue_s = [&]() __attribute__((always_inline)) {
                    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(sring + s_issue * S_UNITS + wv * 64);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(vs), "s"(ssrc), "s"(dst) : "memory");
                    ssrc += 4096;
                    s_issue ^= 1;
                };
                constexpr bool DEPTH1 = (CHASE_M3_DEPTH == 1);
                const int npre = min(nfull, DEPTH1 ? 2 : C_::STAGES);
                issue_s(); issue();                                            // S(0), T(0)
                if (npre > 1) { issue_s(); issue(); }                          // S(1), T(1)
                if (npre > 2) issue();                                         // T(2)
                if (npre == 3)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G + 1) : "memory");
                else if (npre == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G + 1) : "memory");
                else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                auto ld_a = [&](int stage, int ch, int i) __attribute__((always_inline)) -> d2_t {
                    const d2_t* sA = lds + stage * C_::A_UNITS;
                    const int ku = 4 * ch + q, r = wrow + 16 * i + c16;
                    if constexpr (OPA_C) return sA[kidx(r, ku)];
                    else                 return sA[ku * UM + r];
                };
                auto ld_b = [&](int stage, int ch, int j) __attribute__((always_inline)) -> d2_t {
                    const d2_t* sB = lds + C_::STAGES * C_::A_UNITS + stage * C_::B_UNITS;
                    return sB[kidx(wcol + 16 * j + c16, 4 * ch + q)];
                };
                // sum fragment of column tile j of chunk ch (k = 4 ch + q, column 16 j + c16) in the ring stage `soff` points into:
                // soff = this lane's byte offset of the ring stage being READ NEXT, toggled (one 32-bit xor) at every mid-step
                // barrier - ring base 72 KB and stage size 4 KB: bit 12 selects the stage, the lane part stays below 2 KB
                static_assert(((C_::STAGES * C_::STAGE_UNITS * 16) & 4096) == 0, "ring base must have bit 12 clear");
                // image of a ring stage: [4 k pairs][64 columns][k even, k odd] doubles - the 32 lanes (c16, q = 0 / 1) of one half
                // of a ds_read_b64 then read 256 contiguous bytes (all 64 banks once; the plain [8 k][64 columns] image was a
                // two-way conflict, SQ_LDS_BANK_CONFLICT 1.07e10 cycles per launch)
                unsigned soff = (unsigned)(C_::STAGES * C_::STAGE_UNITS * 16) + (unsigned)((q >> 1) * 1024 + c16 * 16 + (q & 1) * 8);
                auto ld_s = [&](int ch, int j) __attribute__((always_inline)) -> double {
                    return *(const double*)((const char*)lds + soff + ch * 2048 + j * 256);
                };
                d2_t aC[TM], bF[TN];
                double sS[TN];
                #pragma unroll
                for (int i = 0; i < TM; ++i) aC[i] = ld_a(0, 0, i);
                #pragma unroll
                for (int j = 0; j < TN; ++j) { bF[j] = ld_b(0, 0, j); sS[j] = ld_s(0, j); }
                // One MFMA cluster on (aC, bF, sS) while the fragments of (nstage, nch) stream in; `more`: there is a next
                // chunk.  fill_t: the cluster also requests tile kt+STAGES (its six copies, one per MFMA group, slots 1..6);
                // fill_s: and first the plane block two K steps ahead (slot 0).
                auto cluster = [&](int nstage, int nch, auto more_c, auto fill_t_c, auto fill_s_c, int fstage = 0) __attribute__((always_inline)) {
                    constexpr bool more = decltype(more_c)::value;
                    constexpr bool fill_t = decltype(fill_t_c)::value;
                    constexpr bool fill_s = decltype(fill_s_c)::value;
                    #pragma unroll
                    for (int i = 0; i < TM; ++i) {
#ifdef CHASE_DIAG_NO_SA
                        // diagnostic build (timing only, results wrong on purpose): what would a launch gain if the H-side operand sums
                        // cost nothing (profiles/r06_dropped_with_data.txt)
                        const double sa = aC[i].x;
#else
                        const double sa = OPA_C ? aC[i].x - aC[i].y : aC[i].x + aC[i].y;
#endif
                        #pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (!RAGGED || j < jv) {                           // ragged tile: groups past n carry no MFMAs
                                acc[0][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bF[j].x, aC[i].x, acc[0][j][i], 0, 0, 0);
                                acc[1][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bF[j].y, aC[i].y, acc[1][j][i], 0, 0, 0);
                                acc[2][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(sS[j], sa, acc[2][j][i], 0, 0, 0);
                            }
                            const int slot = TN * i + j;
                            if (fill_s && slot == 0) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_s();
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (fill_t && slot >= 1 && slot - 1 < NA + NB) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_one(slot - 1, fstage);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            // (the barrier keeps the refills BEHIND their registers' last readers: hoisted above them they
                            // would need registers of their own - there are none)
                            __builtin_amdgcn_sched_barrier(0);
                            if (more && i == TM - 1) { bF[j] = ld_b(nstage, nch, j); sS[j] = ld_s(nch, j); }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (more) aC[i] = ld_a(nstage, nch, i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                static_assert(TM * TN >= NA + NB + 1, "one copy per (i, j) slot of the cluster");
                constexpr std::true_type yes{};
                constexpr std::false_type no{};
                // one K step on the tile in stage st; fill_t: tile kt+STAGES exists and goes into stage st; fill_s: tile kt+2
                // exists, its plane block goes into the ring stage this step has just finished reading
                auto kstep = [&](int kt, int st, auto fill_t_c, auto fill_s_c) __attribute__((always_inline)) {
                    const int stn = (st + 1 == C_::STAGES) ? 0 : st + 1;
                    cluster(st, 1, yes, no, no);                               // chunk 0 of tile kt, prefetching its chunk 1
                    // my reads of stage st and of its ring stage are complete, my copies of tile kt+1 and of its plane block
                    // have landed; tile kt+2 (requested one step ago) may stay in flight
                    if (!DEPTH1 && (decltype(fill_t_c)::value || kt + 2 < nfull))
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    soff ^= 4096u;                                             // the other ring stage: tile kt+1's sums
                    __builtin_amdgcn_sched_barrier(0);
                    // (DEPTH1: the tile requested here is kt+2, into the stage BEHIND st, free since the previous step)
                    cluster(stn, 0, yes, fill_t_c, fill_s_c, DEPTH1 ? (st == 0 ? C_::STAGES - 1 : st - 1) : st);   // chunk 1 of tile kt (+ the copies)
                };
                int kt = 0;
                // steady state, three K steps per trip: the stage indices are compile-time constants, so every LDS address is a
                // loop-invariant register plus an immediate offset (the ring stage of the sums: one xor per K step)
                constexpr int AHEAD = DEPTH1 ? 2 : C_::STAGES;                 // the tile a step requests: kt + AHEAD
                if constexpr (C_::STAGES == 3) {
                    for (; kt + 2 + AHEAD < nfull; kt += 3) {
                        kstep(kt, 0, yes, yes);
                        kstep(kt + 1, 1, yes, yes);
                        kstep(kt + 2, 2, yes, yes);
                    }
                }
                int st = 0;                                                    // kt is a multiple of STAGES here
                for (; kt + AHEAD < nfull; ++kt) { kstep(kt, st, yes, yes); st = (st + 1 == C_::STAGES) ? 0 : st + 1; }
                if (!DEPTH1 && kt + 2 < nfull)        { kstep(kt, st, no, yes);  st = (st + 1 == C_::STAGES) ? 0 : st + 1; ++kt; }
                for (; kt + 1 < nfull; ++kt)          { kstep(kt, st, no, no);   st = (st + 1 == C_::STAGES) ? 0 : st + 1; }
                cluster(st, 1, yes, no, no);                                   // last tile: nothing to publish or prefetch after it
                __builtin_amdgcn_sched_barrier(0);
                cluster(st, 1, no, no, no);
            } else if constexpr (CPLX && M3 && CHASE_M3_PIPELINE) {
                // (rounds 2-5, kept for comparison builds: -DCHASE_M3_SPLANE=0)
                // 3M software pipeline.  192 accumulator registers leave no room for a second set of fragments, so only
                // the A fragments (8 registers) are double-buffered; each B fragment is REFILLED IN PLACE with the next
                // chunk's data right after the six MFMAs that were its last readers, i.e. 18 MFMAs (>1000 cycles) before
                // it is needed again.  The barrier that publishes tile kt+1 sits between the two MFMA clusters like in the
                // four-product loop above, and the stage of tile kt is refilled as soon as its last fragment has been read.
                constexpr int G = C_::GLDS_PER_WAVE;
                const int npre = min(nfull, C_::STAGES);
                for (int t = 0; t < npre; ++t) issue();
                if (npre == 3)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
                else if (npre == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
                else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                auto ld_a = [&](int stage, int ch, d2_t (&a)[TM]) __attribute__((always_inline)) {
                    const d2_t* sA = lds + stage * C_::STAGE_UNITS;
                    const int ku = 4 * ch + q;
                    #pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int r = wrow + 16 * i + c16;
                        if constexpr (OPA_C) a[i] = sA[kidx(r, ku)];
                        else                 a[i] = sA[ku * UM + r];
                    }
                };
                auto ld_b = [&](int stage, int ch, int j) __attribute__((always_inline)) -> d2_t {
                    const d2_t* sB = lds + stage * C_::STAGE_UNITS + C_::A_UNITS;
                    return sB[kidx(wcol + 16 * j + c16, 4 * ch + q)];
                };
                d2_t aC[TM], aN[TM], bF[TN];
                ld_a(0, 0, aC);
                #pragma unroll
                for (int j = 0; j < TN; ++j) bF[j] = ld_b(0, 0, j);
                // One MFMA cluster on (aC, bF) while the fragments of (nstage, nch) stream in; `more`: there is a next chunk.
                // `fill`: the cluster also requests tile kt+STAGES: its six global -> LDS copies are spread over the cluster,
                // one after every third MFMA, so that each copy's issue slot lies under an MFMA in flight (issued as one
                // block after the barrier they cost 4 % of the loop, profiles/r02_mfma_f64_issue.txt).
                auto cluster = [&](int nstage, int nch, auto more_c, auto fill_c, int fstage = 0) __attribute__((always_inline)) {
                    constexpr bool more = decltype(more_c)::value;
                    constexpr bool fill = decltype(fill_c)::value;
                    double sa[TM];
                    #pragma unroll
                    for (int i = 0; i < TM; ++i) sa[i] = OPA_C ? aC[i].x - aC[i].y : aC[i].x + aC[i].y;
                    if (more) ld_a(nstage, nch, aN);
                    #pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const double sb = bF[j].x + bF[j].y;
                        #pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            if (!RAGGED || j < jv) {                           // ragged tile: groups past n carry no MFMAs
                                acc[0][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bF[j].x, aC[i].x, acc[0][j][i], 0, 0, 0);
                                acc[1][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bF[j].y, aC[i].y, acc[1][j][i], 0, 0, 0);
                                acc[2][j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(sb, sa[i], acc[2][j][i], 0, 0, 0);
                            }
                            if (fill && (TM * j + i) < NA + NB) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_one(TM * j + i, fstage);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        if (more) bF[j] = ld_b(nstage, nch, j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    #pragma unroll
                    for (int i = 0; i < TM; ++i) aC[i] = aN[i];
                };
                static_assert(TM * TN >= NA + NB, "one copy per (j, i) slot of the cluster");
                constexpr std::true_type yes{};
                constexpr std::false_type no{};
                // one K step on the tile in stage st; `fill`: tile kt+STAGES exists and goes into stage st
                auto kstep = [&](int kt, int st, auto fill_c) __attribute__((always_inline)) {
                    const int stn = (st + 1 == C_::STAGES) ? 0 : st + 1;
                    cluster(st, 1, yes, no);                                   // chunk 0 of tile kt, prefetching its chunk 1
                    // my reads of stage st are complete and my copies of tile kt+1 have landed; tile kt+2 (requested one
                    // step ago) may stay in flight
                    if (C_::STAGES > 2 && (decltype(fill_c)::value || kt + 2 < nfull))
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    cluster(stn, 0, yes, fill_c, st);                          // chunk 1 of tile kt (+ the copies of tile kt+STAGES)
                };
                int kt = 0;
                // steady state, three K steps per trip: the stage indices are compile-time constants, so every LDS address is a
                // loop-invariant register plus an immediate offset (no address arithmetic in the loop).  In the two ragged op = C
                // instantiations this form parks ONE VGPR in scratch across the K loop (one store before the loops, one load after
                // them - tests/test_kernel_resources.py checks that no scratch access sits inside a loop); the form without the
                // unrolled loop needs no scratch but runs all-ragged launches 3 % slower (profiles/r04_ragged_c_compare.txt), so the
                // unrolled form stays.  -DCHASE_NO_RAGGED_C_UNROLL builds the comparison.
#ifdef CHASE_NO_RAGGED_C_UNROLL
                constexpr bool unroll3 = (C_::STAGES == 3) && !(RAGGED && OPA_C);
#else
                constexpr bool unroll3 = (C_::STAGES == 3);
#endif
                if constexpr (unroll3) {
                    for (; kt + 2 + C_::STAGES < nfull; kt += 3) {
                        kstep(kt, 0, yes);
                        kstep(kt + 1, 1, yes);
                        kstep(kt + 2, 2, yes);
                    }
                }
                int st = 0;                                                    // kt is a multiple of STAGES here
                for (; kt + C_::STAGES < nfull; ++kt) { kstep(kt, st, yes); st = (st + 1 == C_::STAGES) ? 0 : st + 1; }
                for (; kt + 1 < nfull; ++kt)          { kstep(kt, st, no);  st = (st + 1 == C_::STAGES) ? 0 : st + 1; }
                cluster(st, 1, yes, no);                                       // last tile: nothing to publish or prefetch after it
                __builtin_amdgcn_sched_barrier(0);
                cluster(st, 1, no, no);
            } else {
                // real (80 fragment registers next to 128 accumulator registers) cannot double-buffer fragments: one barrier
                // per K step, each chunk's fragments fetched right before its MFMAs (also the 3M loop without pipelining)
                constexpr int DEPTH = C_::STAGES - 1;
                issue();
                if (DEPTH > 1 && nfull > 1) issue();
                int st_comp = 0;
                // the copies of tile kt+DEPTH go out under the first MFMA groups of tile kt's first cluster (one copy per
                // group), into the stage whose last reader passed the barrier of this step
                auto kstep = [&](int kt, auto fill_c) __attribute__((always_inline)) {
                    // my own copies of tile kt have landed (with three stages tile kt+1 may stay in flight) ...
                    if (DEPTH > 1 && kt + 1 < nfull) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C_::GLDS_PER_WAVE) : "memory");
                    else                             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    // ... and after the barrier everybody's have; everybody has also finished reading the stage refilled next
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    const int st_fill = (st_comp + DEPTH) % C_::STAGES;
                    compute(st_comp, [&](int slot) __attribute__((always_inline)) {
                        if (decltype(fill_c)::value && slot < NA + NB) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_one(slot, st_fill);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    st_comp = (st_comp + 1 == C_::STAGES) ? 0 : st_comp + 1;
                };
                static_assert(2 * TN >= NA + NB, "one copy per MFMA group of the cluster");
                int kt = 0;
                for (; kt + DEPTH < nfull; ++kt) kstep(kt, std::true_type());
                for (; kt < nfull; ++kt)         kstep(kt, std::false_type());
            }
            if constexpr (!M3) {
            // a partial last K tile goes through the guarded register path into the stage nobody reads any more
            if (nfull * BK < kend - kbeg) {
                const int st = nfull % C_::STAGES;
                __syncthreads();
                load_tile(kbeg + nfull * BK);
                store_tile(st);
                __syncthreads();
                compute(st, no_hook);
            }
            }
            done = true;
        }
    }
    if constexpr (!M3) {      // M3 launches are only made when every workgroup takes the LDS-DMA path (host check)
    if (!done && nkt > 0) {
        load_tile(kbeg);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) load_tile(kbeg + (kt + 1) * BK);
            compute(cur, no_hook);
            if (kt + 1 < nkt) store_tile(cur ^ 1);
            __syncthreads();
        }
    }

    }
    // ---- epilogue -----------------------------------------------------------------------------------------------
    // accumulator element: D[row = q + 4g -> N index][col = c16 -> M index]
    // raw blocks store the unscaled partial tile into their slab (tile-local coordinates, ld = BM, no bounds)
    double* Cb = raw ? slab : p.C;
    const long ldc_e = raw ? BM : p.ldc;
    const int srow0 = raw ? 0 : row0, scol0 = raw ? 0 : col0;
    const int m_e = raw ? BM : p.m, n_e = raw ? BN : p.n;
    const double are = p.alpha_re, aim = p.alpha_im, bre = p.beta_re, bim = p.beta_im;
    const bool has_beta = (bre != 0.0) || (bim != 0.0);

    #pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (j >= jv) continue;                      // nothing accumulated there (tail_reduce never reads those columns)
        #pragma unroll
        for (int g = 0; g < 4; ++g) {
            // keep the C reads of one accumulator register group together: hoisting all 32 groups' loads above the stores
            // would need more registers than the accumulators leave
            __builtin_amdgcn_sched_barrier(0);
            const int gj = scol0 + wcol + 16 * j + q + 4 * g;
            if (gj >= n_e) continue;
            if constexpr (CPLX) {
                #pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int gi = srow0 + wrow + 16 * i + c16;
                    if (gi >= m_e) continue;
                    double* c = Cb + ((long)gj * ldc_e + gi) * 2;
                    double xr, xi;
                    if constexpr (M3) {
                        const double p1 = acc[0][j][i][g], p2 = acc[1][j][i][g], p3 = acc[NACC - 1][j][i][g];
                        xr = OPA_C ? p1 + p2 : p1 - p2;
                        xi = OPA_C ? (p3 - p1) + p2 : (p3 - p1) - p2;
                    } else { xr = acc[0][j][i][g]; xi = acc[1][j][i][g]; }
                    d2_t out;
                    if (raw) { out = d2_t{xr, xi}; }
                    else {
                        out = d2_t{are * xr - aim * xi, are * xi + aim * xr};
                        if (has_beta) {
                            const d2_t old = *(const d2u_t*)c;
                            out.x += bre * old.x - bim * old.y;
                            out.y += bre * old.y + bim * old.x;
                        }
                    }
                    *(d2u_t*)c = out;
                }
            } else if constexpr (!OPA_C) {
                #pragma unroll
                for (int pr = 0; pr < TM / 2; ++pr) {
                    const int gi = srow0 + wrow + 32 * pr + 2 * c16;         // rows gi, gi+1 <- tiles 2pr, 2pr+1
                    if (gi >= m_e) continue;
                    double* c = Cb + (long)gj * ldc_e + gi;
                    double x0 = acc[0][j][2 * pr][g], x1 = acc[0][j][2 * pr + 1][g];
                    if (!raw) {
                        x0 *= are; x1 *= are;
                        if (has_beta) { x0 += bre * c[0]; if (gi + 1 < m_e) x1 += bre * c[1]; }
                    }
                    if (gi + 1 < m_e) *(d2u_t*)c = d2_t{x0, x1};
                    else c[0] = x0;
                }
            } else {
                #pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int gi = srow0 + wrow + 16 * i + c16;
                    if (gi >= m_e) continue;
                    double* c = Cb + (long)gj * ldc_e + gi;
                    double x0 = acc[0][j][i][g];
                    if (!raw) { x0 *= are; if (has_beta) x0 += bre * c[0]; }
                    c[0] = x0;
                }
            }
        }
    }
}

// tail tiles: C[tile] = alpha * sum_z slab[tile][z] + beta * C[tile]   (fixed summation order => bitwise reproducible)
// TAIL_PARTS workgroups per tail tile (each sums its share of the tile's elements over the sk slabs in slab order);
// slabs are BM x BN tiles (ld = BM) in tile-local coordinates
constexpr int TAIL_PARTS = 4;
template <bool CPLX, int BM, int BN>
__global__ __launch_bounds__(256) void tail_reduce_kernel(const double* __restrict__ slabs, int first_tile, int sk,
                                                          int gn, int m, int n, double* __restrict__ C, long ldc,
                                                          double are, double aim, double bre, double bim, int group_rows,
                                                          int bn_cols)
{
    constexpr int EPT = CPLX ? 2 : 1;
    const int tt = blockIdx.x / TAIL_PARTS, part = blockIdx.x % TAIL_PARTS;
    const int tile = first_tile + tt;
    int bm, bn;
    tile_coords(tile, (m + BM - 1) / BM, gn, group_rows, bm, bn);
    const int row0 = bm * BM, col0 = bn * bn_cols;
    const double* base = slabs + (size_t)tt * sk * (BM * BN * EPT);
    const bool has_beta = (bre != 0.0) || (bim != 0.0);
    constexpr int SHARE = BM * BN / TAIL_PARTS;
    for (int e = part * SHARE + threadIdx.x; e < (part + 1) * SHARE; e += 256) {
        const int li = e % BM, lj = e / BM;
        const int gi = row0 + li, gj = col0 + lj;
        if (gi >= m || gj >= n || lj >= bn_cols) continue;      // columns past the tile's stride belong to the next tile
        if constexpr (CPLX) {
            double sr = 0.0, si = 0.0;
            for (int z = 0; z < sk; ++z) {
                const d2_t v = *(const d2_t*)(base + ((size_t)z * BM * BN + e) * 2);
                sr += v.x; si += v.y;
            }
            double* c = C + ((long)gj * ldc + gi) * 2;
            double orr = are * sr - aim * si, oi = are * si + aim * sr;
            if (has_beta) { const double cr = c[0], ci = c[1]; orr += bre * cr - bim * ci; oi += bre * ci + bim * cr; }
            c[0] = orr; c[1] = oi;
        } else {
            double sum = 0.0;
            for (int z = 0; z < sk; ++z) sum += base[(size_t)z * BM * BN + e];
            double* c = C + (long)gj * ldc + gi;
            double o = are * sum;
            if (has_beta) o += bre * c[0];
            c[0] = o;
        }
    }
}

// The plane of V-side operand sums of a 3M launch (GemmArgs::S): for column tile bn (columns col0 = bn * bn_cols .. + 63, clamped
// to the last valid column like the kernel's own B copies) and K step kt the 4 KB block at S + (bn * nkt + kt) * 512, laid out
// [k pair][column][k even, k odd]: element (kk >> 1) * 128 + 2 c + (kk & 1) = re + im of B[k = 8 kt + kk, col0 + c].  One workgroup per (64 k, column tile): B is read along k (1 KB per wave instruction),
// transposed through LDS, written along c (512 B per wave instruction).  2.7 GB read + 1.3 GB written at config 4's full width:
// < 0.1 % of the product it precedes.
__global__ __launch_bounds__(256) void splane_kernel(const double* __restrict__ B, long ldb, int n, int k, int bn_cols, int nkt,
                                                     double* __restrict__ S)
{
    __shared__ double sm[64][65];
    const int bn = blockIdx.y, k0 = blockIdx.x * 64, col0 = bn * bn_cols, t = threadIdx.x;
    #pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int c = 4 * it + (t >> 6), kl = t & 63;
        const int col = min(col0 + c, n - 1);
        double v = 0.0;
        if (k0 + kl < k) { const d2_t b = *(const d2u_t*)(B + ((long)col * ldb + k0 + kl) * 2); v = b.x + b.y; }
        sm[kl][c] = v;
    }
    __syncthreads();
    // one 16-byte store per (k pair, column): a wave writes 1 KB contiguous
    #pragma unroll 4
    for (int it = 0; it < 8; ++it) {
        const int kp = 4 * it + (t >> 6), c = t & 63, kk = k0 + 2 * kp;          // k is a multiple of 8 for every 3M launch
        if (kk < k) *(d2_t*)(S + ((size_t)bn * nkt + (kk >> 3)) * 512 + ((kk & 7) >> 1) * 128 + c * 2) = d2_t{sm[2 * kp][c], sm[2 * kp + 1][c]};
    }
}
// bytes of the plane for an m x n x k piece in column tiles of bn_cols (0: whole tiles)
static size_t splane_bytes(int n, int k, int bn_cols)
{
    if (bn_cols <= 0 || bn_cols > 64) bn_cols = 64;
    return (size_t)((n + bn_cols - 1) / bn_cols) * (size_t)((k + 7) / 8) * 4096;
}

// three-multiplication scheme for the complex filter products: -1 = not decided yet (CHASE_HIP_GEMM3M, default on)
static std::atomic<int> g_gemm3m{-1};
int gemm3m_enabled()
{
    int v = g_gemm3m.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("CHASE_HIP_GEMM3M");
        v = (e ? atoi(e) != 0 : true) ? 1 : 0;
        g_gemm3m.store(v, std::memory_order_relaxed);
    }
    return v;
}
void gemm3m_set(int on) { g_gemm3m.store(on ? 1 : 0, std::memory_order_relaxed); }

// split-K factor of the tail tiles (see GemmArgs): minimises rounds of the chip, capped by the slabs the workspace holds
static int choose_tail_sk(long tail, long slots, int nkt, size_t slab_bytes, size_t ws_bytes)
{
    if (tail <= 0 || nkt < 16) return 1;
    const int sk_max = (int)std::min<long>(std::min<long>(64, nkt / 8), (long)(ws_bytes / (slab_bytes * (size_t)tail)));
    int sk = 1;
    double best = 1e30;
    for (int c = 1; c <= sk_max; ++c) {
        const double rounds = (double)((tail * c + slots - 1) / slots) / c + 0.004 * c;
        if (rounds < best - 1e-12) { best = rounds; sk = c; }
    }
    return sk;
}

// per-device attribute flags, executed-flop accounting; min_rounds > 0: a product with fewer than min_rounds tiles per
// workgroup slot is cut along K until it has that many pieces per slot (see forced_split)
struct LaunchInfo { int device; double* exec_flops; int min_rounds; };

// Granularity for launches that SHARE the chip with a collective (the panel products of the pipelined distributed HEMM): a
// panel is sized to fill the 512 workgroup slots exactly once, and the kernel's two workgroups per CU leave no registers or
// LDS for another kernel's waves, so an all-reduce kernel that needs a few CUs displaces whole tiles into a second round -
// the panel would take twice as long.  Cut into `sk` K pieces per tile (raw slabs + the fixed-order reduction of the tail
// path), the displaced work is a fraction of a tile.  Returns the split (1: leave the automatic decomposition alone).
static int forced_split(long tiles, long slots, int nkt, size_t slab_bytes, int min_rounds)
{
    if (min_rounds <= 0 || tiles <= 0 || tiles >= (long)min_rounds * slots) return 1;
    long sk = ((long)min_rounds * slots + tiles - 1) / tiles;
    sk = std::min<long>(sk, 8);
    sk = std::min<long>(sk, nkt / 64);                                        // pieces of at least 64 K steps
    sk = std::min<long>(sk, (long)(GEMM_WS_CAP / (slab_bytes * (size_t)tiles)));
    return sk < 2 ? 1 : (int)sk;
}
constexpr int MAX_DEVICES = 64;

// The decomposition of ONE launch (m x n x k in column tiles of bn_cols columns): whole tiles, tail tiles cut into sk K
// pieces, the slab bytes that needs.  A function of the shape, the chip and min_rounds ONLY - the split, and with it the
// summation order, never depends on how far a caller's workspace happens to have grown.  The one place both the launcher
// (launch_gemm_part) and the workspace sizing (gemm_f64_ws_need) take it from.
struct PartPlan { long full, tail; int sk; size_t ws_bytes; };
template <bool CPLX, bool OPA_C, bool NARROW = false>
static PartPlan plan_part(int m, int n, int k, int bn_cols, int num_cu, int min_rounds)
{
    using C_ = Cfg<CPLX, OPA_C, NARROW>;
    if (bn_cols <= 0 || bn_cols > C_::BN) bn_cols = C_::BN;
    const long tiles = (long)((m + C_::BM - 1) / C_::BM) * ((n + bn_cols - 1) / bn_cols);
    const long slots = 2L * num_cu;                          // two workgroups per CU
    const int nkt = (k + C_::BK - 1) / C_::BK;
    const size_t slab_bytes = (size_t)C_::BM * C_::BN * sizeof(double) * C_::EPT;
    PartPlan p{tiles, 0, 1, 0};
    if (m <= 0 || n <= 0 || k <= 0) return p;
    const int fs = forced_split(tiles, slots, nkt, slab_bytes, min_rounds);
    if (fs > 1) { p.full = 0; p.tail = tiles; p.sk = fs; }  // shared-chip launch: every tile in K pieces
    else {
        // cost of the tail in units of "one tile on one slot": ceil(tail*sk/slots)/sk rounds, slightly penalising big sk
        const long tail = tiles % slots;
        const int sk = tail > 0 ? choose_tail_sk(tail, slots, nkt, slab_bytes, GEMM_WS_CAP) : 1;
        if (sk > 1) { p.full = tiles - tail; p.tail = tail; p.sk = sk; }      // else: every tile is a whole tile
    }
    p.ws_bytes = p.tail > 0 ? slab_bytes * (size_t)p.tail * p.sk : 0;
    return p;
}

// How a width n is covered by launches (launch_gemm_cols): at most two pieces of columns, each one launch of tiles that all
// have the same stride (bn_cols = 0: the tile width; 16 g: "uniform ragged" tiles of g 16-column groups) - a launch never mixes
// whole and partial tiles (see launch_gemm_cols).  `narrow`: the piece runs on the 128 x 64 real tile.
struct ColPiece { int c0, n, bn_cols; bool narrow; };
struct ColsPlan { int npieces; ColPiece piece[2]; };
template <bool CPLX, bool OPA_C> static int uniform_tile_cols(int m, int n, int k);
static int real_narrow_mode()
{
    // 1 (default): narrow tiles for blocks of at most 64 columns and for the ragged rest of a width; 0: 128-wide tiles only
    static const int mode = [] { const char* e = getenv("CHASE_HIP_REAL_NARROW"); return e ? atoi(e) : 1; }();
    return mode;
}
template <bool CPLX, bool OPA_C>
static ColsPlan plan_cols(int m, int n, int k, bool have_ws)
{
    using C_ = Cfg<CPLX, OPA_C>;
    const ColsPlan single{1, {{0, n, 0, false}, {0, 0, 0, false}}};
    const int rem = n % C_::BN;
    if constexpr (!CPLX) {
        const int mode = real_narrow_mode();
        // uniform narrow tiles covering r columns: as few 64-wide tiles as hold its 16-column groups, the same number of
        // groups in each
        auto narrow_stride = [](int r) { const int ng = (r + 15) / 16, nt = (ng + 3) / 4; return 16 * ((ng + nt - 1) / nt); };
        if (mode != 0 && n <= 64) return ColsPlan{1, {{0, n, narrow_stride(n), true}, {0, 0, 0, false}}};
        if (!have_ws || n <= C_::BN || rem == 0) return single;
        // The ragged rest of a width gets a launch of its own (a launch never mixes whole and partial tiles), K-split over
        // the chip: up to 64 columns on ONE column of narrow tiles - measured at N = 32768 (profiles/r03_hemm_sweep.txt):
        // 1.48 ms for one 16-column group (a pure stream over A at 5.8 TB/s), 2.0 ms for three, 2.4 ms for four, against
        // 1.9 / 2.4 / 2.9 ms on the 128-wide tile with its two LDS stages; two columns of narrow tiles would stream A twice
        // (4.2 ms for 72 columns against 3.3 ms), so a wider rest stays on the 128-wide ragged launch.  A rest of more than
        // 112 columns fills its last tile well enough to stay in the one launch.
        if (rem > C_::BN - 16) return single;
        if (mode != 0 && rem <= 64) {
            // cheaper than letting a partly idle 128-wide last tile ride along at a whole tile's cost? (fit of the numbers
            // above: 0.75 passes over A + 0.68 of a group's whole-tile time per live group)
            // (75 TFLOP/s and 5.5 TB/s are FIXED model constants, not read from the device: the choice sets the launch
            // decomposition and with it the summation order, which must not differ between the devices of one job)
            const double t_group = 2.0 * m * (double)k * 16.0 / 75.0e12;
            const double t_pass = (double)m * k * 8.0 / 5.5e12;
            const int g = (rem + 15) / 16;
            if (mode == 2 || 0.75 * t_pass + 0.68 * g * t_group + 10e-6 < C_::TN * t_group)
                return ColsPlan{2, {{0, n - rem, 0, false}, {n - rem, rem, 16 * g, true}}};
            return single;
        }
        return ColsPlan{2, {{0, n - rem, 0, false}, {n - rem, rem, 0, false}}};
    } else {
        const bool balanced = (C_::WAVES_N == 1);      // all waves span the tile width: skipped groups cost nobody
        if (!(have_ws && balanced && n > C_::BN && rem != 0 && rem <= C_::BN - 16)) return single;
        const int bnu = uniform_tile_cols<CPLX, OPA_C>(m, n, k);
        if (bnu > 0) return ColsPlan{1, {{0, n, bnu, false}, {0, 0, 0, false}}};   // every tile the same number of groups
        return ColsPlan{2, {{0, n - rem, 0, false}, {n - rem, rem, 0, false}}};
    }
}

template <bool CPLX, bool OPA_C, int TAG, bool NARROW = false>
static int launch_gemm_part(hipStream_t st, int m, int n, int k, const double* alpha, const double* A, long lda,
                            const double* B, long ldb, const double* beta, double* C, long ldc,
                            double* ws, size_t ws_bytes, int num_cu, bool allow3m, const LaunchInfo& li,
                            int bn_cols = 0)
{
    using C_ = Cfg<CPLX, OPA_C, NARROW>;
    if (m <= 0 || n <= 0) return 0;
    if (bn_cols <= 0 || bn_cols > C_::BN) bn_cols = C_::BN;
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.m = m; a.n = n; a.k = k;
    a.gm = (m + C_::BM - 1) / C_::BM; a.gn = (n + bn_cols - 1) / bn_cols;
    a.bn_cols = bn_cols;
    // Row panels per group of the tile order (tile_coords): 2; FOUR for launches of many rounds with many column tiles (round 6).
    // With the plane-fed 3M loop the workgroups that share panels stay in step, so a 4 x 16 patch of tiles per XCD (256 KB of
    // unique operands per K step instead of 416 KB) really is served by the L2: 0.89 instead of 1.31 TB through the fabric for
    // the full-width config-4 launch, the same time on a device that holds its clock (873 vs 875 ms) and 2 % less on one that
    // does not under this load (877 vs 892-898 ms: profiles/r06_tile_group.txt).  Short launches and the 256-column panels of the
    // grid pipeline (4 column tiles) are 2.5 % SLOWER with 4, hence the rule - a function of the shape alone (the order decides
    // which tiles form the K-split tail, i.e. the summation order, which must not depend on the device).
    // CHASE_HIP_TILE_GROUP=<n> fixes it for every launch.
    static const int group_env = [] { const char* e = getenv("CHASE_HIP_TILE_GROUP"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : v; }();
    a.group_rows = group_env > 0 ? group_env : ((a.gn >= 16 && (long)a.gm * a.gn >= 16L * 2 * num_cu) ? 4 : 2);
    a.alpha_re = alpha[0]; a.alpha_im = CPLX ? alpha[1] : 0.0;
    a.beta_re = beta[0];   a.beta_im = CPLX ? beta[1] : 0.0;
    const int nkt = (k + C_::BK - 1) / C_::BK;
    // without a workspace nothing can be split; with one, the plan's slabs must fit - callers size the workspace with
    // gemm_f64_ws_need, which takes its numbers from the same plan
    PartPlan pl = plan_part<CPLX, OPA_C, NARROW>(m, n, k, bn_cols, num_cu, li.min_rounds);
    if (ws == nullptr) pl = PartPlan{(long)a.gm * a.gn, 0, 1, 0};
    if (pl.ws_bytes > ws_bytes) return GEMM_F64_EWORKSPACE;             // never split differently to fit: refuse (distinct code)
    const long full = pl.full, tail = pl.tail;
    const int sk = pl.sk;
    int kchunk = ((nkt + sk - 1) / sk) * C_::BK;
    a.full_tiles = (int)full; a.tail_sk = sk; a.tail_kchunk = kchunk; a.slabs = ws;
    // global_load_lds moves 16 bytes per lane: complex elements always qualify, real ones need even leading dimensions
    a.glds_ok = (((uintptr_t)A | (uintptr_t)B) % 16 == 0) && (CPLX || ((lda % 2 == 0) && (ldb % 2 == 0))) &&
                (double)std::max(lda, ldb) * C_::BM * C_::EPT * 8 < 4.0e9;      // per-lane byte offsets are 32-bit
    const unsigned grid = (unsigned)(full + tail * sk);
    const size_t lds_bytes = (size_t)C_::STAGES * C_::STAGE_UNITS * sizeof(d2_t);
    // 3M kernels: + the ring of two 4 KB stages of the operand-sum plane (80 KB: two workgroups per CU still fit)
    const size_t lds_bytes3 = lds_bytes + (CHASE_M3_SPLANE ? 2 * 4096 : 0);
    // ragged: some 16-column group of the last column tile lies entirely past n
    const bool ragged = (bn_cols < C_::BN) || (a.gn * C_::BN - n) >= 16;
    // the dynamic-LDS limit is a per-device function attribute: one flag per device (setting it twice is harmless, so a
    // relaxed atomic is enough for concurrent first calls)
    const int dev = (li.device >= 0 && li.device < MAX_DEVICES) ? li.device : 0;
    static std::atomic<bool> attr_set[MAX_DEVICES];
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        (void)hipFuncSetAttribute((const void*)gemm_f64_kernel<CPLX, OPA_C, TAG, false, false, NARROW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute((const void*)gemm_f64_kernel<CPLX, OPA_C, TAG, true, false, NARROW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    // complex HEMMs of the filter phase (tag 1) and the H-times-block products of Rayleigh-Ritz / residuals (tag 2, see
    // gemm_f64) run the 3-product scheme unless CHASE_HIP_GEMM3M=0 / gemm3m_set(0); everything else - Gram products,
    // back-transforms, QR, verification products (tag 3) - stays on four products like the reference's zgemm
    constexpr bool CAN3M = CPLX;
    const bool want3m = gemm3m_enabled() != 0;
    // the 3M instantiation has no register-staged fallback: whole row tiles, whole K tiles, 16-byte addressable operands
    bool ok3m = allow3m && want3m && a.glds_ok && (m % C_::BM == 0) && (k % C_::BK == 0) && (kchunk % C_::BK == 0);
    a.S = nullptr; a.s_nkt = 0;
    if constexpr (CAN3M) {
        static std::atomic<bool> attr3[MAX_DEVICES];
        if (!attr3[dev].load(std::memory_order_relaxed)) {
            (void)hipFuncSetAttribute((const void*)gemm_f64_kernel<CPLX, OPA_C, TAG, false, CAN3M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes3);
            (void)hipFuncSetAttribute((const void*)gemm_f64_kernel<CPLX, OPA_C, TAG, true, CAN3M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes3);
            attr3[dev].store(true, std::memory_order_relaxed);
        }
        if (ok3m && CHASE_M3_SPLANE) {
            // the operand-sum plane sits behind the slabs in the caller's workspace (gemm_f64_ws_need counts it); written on the
            // same stream right before the product
            const size_t off = (pl.ws_bytes + 255) & ~(size_t)255, sb = splane_bytes(n, k, bn_cols);
            if (ws == nullptr) ok3m = false;                                   // no workspace: four products
            else if (off + sb > ws_bytes) return GEMM_F64_EWORKSPACE;
            else {
                double* S = (double*)((char*)ws + off);
                hipLaunchKernelGGL(splane_kernel, dim3((unsigned)((k + 63) / 64), (unsigned)a.gn), dim3(256), 0, st, B, ldb, n, k,
                                   bn_cols, k / C_::BK, S);
                a.S = S; a.s_nkt = k / C_::BK;
            }
        }
        if (ok3m) {
            if (ragged) hipLaunchKernelGGL((gemm_f64_kernel<CPLX, OPA_C, TAG, true, CAN3M>), dim3(grid), dim3(256), lds_bytes3, st, a);
            else        hipLaunchKernelGGL((gemm_f64_kernel<CPLX, OPA_C, TAG, false, CAN3M>), dim3(grid), dim3(256), lds_bytes3, st, a);
        }
    }
    if (!(CAN3M && ok3m)) {
        if (ragged) hipLaunchKernelGGL((gemm_f64_kernel<CPLX, OPA_C, TAG, true, false, NARROW>), dim3(grid), dim3(256), lds_bytes, st, a);
        else        hipLaunchKernelGGL((gemm_f64_kernel<CPLX, OPA_C, TAG, false, false, NARROW>), dim3(grid), dim3(256), lds_bytes, st, a);
    }
    if (tail > 0) {
        hipLaunchKernelGGL((tail_reduce_kernel<CPLX, C_::BM, C_::BN>), dim3((unsigned)tail * TAIL_PARTS), dim3(256), 0, st, ws,
                           (int)full, sk, a.gn, m, n, C, ldc, a.alpha_re, a.alpha_im, a.beta_re, a.beta_im, a.group_rows, a.bn_cols);
    }
    // flops the matrix cores execute for this product: the reference's model 2*F*m*n*k (F = 4 complex), 3/4 of it in 3M
    if (li.exec_flops) *li.exec_flops += 2.0 * (CPLX ? 4.0 : 1.0) * m * (double)n * k * ((CAN3M && ok3m) ? 0.75 : 1.0);
    return (int)hipGetLastError();
}

// "Uniform ragged" tiling of a width that is not a multiple of the tile width: instead of whole tiles + one ragged tile in
// a second launch (which streams A once more - HBM-bound for a handful of columns), give EVERY column tile the same number
// g < TN of 16-column groups (n = 133 complex: 3 tiles of 48 columns instead of 64 + 64 + 5).  All workgroups then skip the
// same groups, so the matrix cores pay for ntiles * g groups and A is streamed once.  Chosen when the model says it is
// cheaper: time per 16-column group = MFMA flops of the group at the measured kernel rate, time of a pass over A = its
// bytes at the measured streaming rate.  Returns the tile stride in columns (0: keep whole tiles + ragged launch).
template <bool CPLX, bool OPA_C>
static int uniform_tile_cols(int m, int n, int k)
{
    using C_ = Cfg<CPLX, OPA_C>;
    static const int mode = [] { const char* e = getenv("CHASE_HIP_UNIFORM_TILES"); return e ? atoi(e) : 1; }();
    if (mode == 0 || (!CPLX && mode != 2)) return 0;      // measured (profiles/r02_hemm_sweep.txt): a gain for the complex kernels only
    const int ng = (n + 15) / 16, ntiles = (ng + C_::TN - 1) / C_::TN, g = (ng + ntiles - 1) / ntiles;
    if (g >= C_::TN) return 0;
    // measured (profiles/r02_hemm_sweep.txt, r02_mfma_f64_issue.txt): ~73 TFLOP/s executed on whole tiles; a launch that only
    // streams A (16 columns) moves it at 6.0 TB/s (N = 32768: 2.87 ms, N = 65536: 11.5 ms)
    const double t_group = 2.0 * (CPLX ? 3.0 : 1.0) * m * (double)k * 16.0 / 73.0e12;
    const double t_pass = (double)m * k * C_::EPT * 8.0 / 6.0e12;
    const int whole = n / C_::BN, rem_groups = (n % C_::BN + 15) / 16;
    const double cost_split = whole * C_::TN * t_group + std::max(rem_groups * t_group, t_pass) + 10e-6;
    const double cost_uniform = std::max((double)ntiles * g * t_group, t_pass);
    if (mode == 2 || cost_uniform < cost_split) return 16 * g;
    return 0;
}

// A ragged last column tile (n % BN != 0) is cheap only next to its own kind: the matrix-core arbiter favours the older
// wave, so a partial-tile workgroup sharing a CU with a whole-tile one advances at the whole tile's pace (measured: no
// gain from skipped MFMAs when mixed).  The ragged column therefore gets its own launch, K-split over the whole chip,
// where every workgroup skips the same 16-column groups.
template <bool CPLX, bool OPA_C, int TAG>
static int launch_gemm_cols(hipStream_t st, int m, int n, int k, const double* alpha, const double* A, long lda,
                            const double* B, long ldb, const double* beta, double* C, long ldc,
                            double* ws, size_t ws_bytes, int num_cu, bool allow3m, const LaunchInfo& li)
{
    constexpr int EPT = CPLX ? 2 : 1;
    const ColsPlan cp = plan_cols<CPLX, OPA_C>(m, n, k, ws != nullptr);
    for (int i = 0; i < cp.npieces; ++i) {
        const ColPiece& pc = cp.piece[i];
        const double* Bp = B + (long)pc.c0 * ldb * EPT;
        double* Cp = C + (long)pc.c0 * ldc * EPT;
        int rc;
        if constexpr (!CPLX) {
            if (pc.narrow) rc = launch_gemm_part<CPLX, OPA_C, TAG, true>(st, m, pc.n, k, alpha, A, lda, Bp, ldb, beta, Cp, ldc, ws, ws_bytes, num_cu, allow3m, li, pc.bn_cols);
            else           rc = launch_gemm_part<CPLX, OPA_C, TAG, false>(st, m, pc.n, k, alpha, A, lda, Bp, ldb, beta, Cp, ldc, ws, ws_bytes, num_cu, allow3m, li, pc.bn_cols);
        } else {
            rc = launch_gemm_part<CPLX, OPA_C, TAG, false>(st, m, pc.n, k, alpha, A, lda, Bp, ldb, beta, Cp, ldc, ws, ws_bytes, num_cu, allow3m, li, pc.bn_cols);
        }
        if (rc) return rc;
    }
    return 0;
}

// The three-multiplication kernel has no guarded path: it needs whole 128-row tiles and whole 8-deep K tiles.  A filter
// product of ARBITRARY size (N = 1001, a local block of 8193 rows ...) is therefore cut into the part that qualifies,
//   C[0:m1, :] = alpha op(A)[0:m1, 0:k1] B[0:k1, :] + beta C[0:m1, :]        m1 = m - m mod 128, k1 = k - k mod 8   (3M)
// plus two thin four-multiplication products for the rims,
//   C[0:m1, :] += alpha op(A)[0:m1, k1:k] B[k1:k, :]                          (< 8 columns of op(A))
//   C[m1:m, :]  = alpha op(A)[m1:m, 0:k]  B + beta C[m1:m, :]                 (< 128 rows)
// instead of running the whole product on four multiplications (25 % more MFMA work) because of a ragged rim.
template <bool CPLX, bool OPA_C>
static bool split3m_applies(int m, int n, int k, bool allow3m)
{
    using C_ = Cfg<CPLX, OPA_C>;
    if (!CPLX || !allow3m || gemm3m_enabled() == 0) return false;
    if (m % C_::BM == 0 && k % C_::BK == 0) return false;              // qualifies as it stands
    return m >= 8 * C_::BM && k >= 64 * C_::BK && n >= 16;             // the rims must be thin next to the bulk
}

template <bool CPLX, bool OPA_C, int TAG>
static int launch_gemm(hipStream_t st, int m, int n, int k, const double* alpha, const double* A, long lda,
                       const double* B, long ldb, const double* beta, double* C, long ldc,
                       double* ws, size_t ws_bytes, int num_cu, bool allow3m, const LaunchInfo& li)
{
    using C_ = Cfg<CPLX, OPA_C>;
    constexpr int EPT = C_::EPT;
    if (!split3m_applies<CPLX, OPA_C>(m, n, k, allow3m))
        return launch_gemm_cols<CPLX, OPA_C, TAG>(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, ws, ws_bytes, num_cu, allow3m, li);
    const int m1 = m - m % C_::BM, k1 = k - k % C_::BK;
    // element (i, kk) of op(A): op = N: A[i + kk lda];  op = C: conj(A[kk + i lda])
    auto Aoff = [&](int i, int kk) { return A + (OPA_C ? ((long)i * lda + kk) : ((long)kk * lda + i)) * EPT; };
    const double one[2] = {1.0, 0.0};
    int rc = launch_gemm_cols<CPLX, OPA_C, TAG>(st, m1, n, k1, alpha, A, lda, B, ldb, beta, C, ldc, ws, ws_bytes, num_cu, true, li);
    if (rc) return rc;
    if (k1 < k) {
        rc = launch_gemm_cols<CPLX, OPA_C, TAG>(st, m1, n, k - k1, alpha, Aoff(0, k1), lda, B + (long)k1 * EPT, ldb, one, C, ldc,
                                                ws, ws_bytes, num_cu, false, li);
        if (rc) return rc;
    }
    if (m1 < m)
        rc = launch_gemm_cols<CPLX, OPA_C, TAG>(st, m - m1, n, k, alpha, Aoff(m1, 0), lda, B, ldb, beta, C + (long)m1 * EPT, ldc,
                                                ws, ws_bytes, num_cu, false, li);
    return rc;
}

// bytes of split-K workspace this product can use (0: none): the slabs of the tail tiles at the split that minimises the
// rounds of the chip (the ragged-column launch is K-split over the whole chip); the caller grows its workspace to this
// on demand instead of holding a fixed allocation
template <bool CPLX, bool OPA_C>
static size_t ws_need(int m, int n, int k, int num_cu, int min_rounds)
{
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const ColsPlan cp = plan_cols<CPLX, OPA_C>(m, n, k, true);
    size_t need = 0;
    for (int i = 0; i < cp.npieces; ++i) {
        const ColPiece& pc = cp.piece[i];
        size_t b;
        if constexpr (!CPLX) {
            b = pc.narrow ? plan_part<CPLX, OPA_C, true>(m, pc.n, k, pc.bn_cols, num_cu, min_rounds).ws_bytes
                          : plan_part<CPLX, OPA_C, false>(m, pc.n, k, pc.bn_cols, num_cu, min_rounds).ws_bytes;
        } else {
            b = plan_part<CPLX, OPA_C, false>(m, pc.n, k, pc.bn_cols, num_cu, min_rounds).ws_bytes;
            // + the operand-sum plane of a 3M launch (whether a product runs on three multiplications depends on the phase
            // tag, which the sizing does not know: counted for every complex product while 3M is enabled)
            if (CHASE_M3_SPLANE && gemm3m_enabled() != 0) b = ((b + 255) & ~(size_t)255) + splane_bytes(pc.n, k, pc.bn_cols);
        }
        need = std::max(need, b);
    }
    return need;
}

size_t gemm_f64_ws_need(bool cplx, char opA, int m, int n, int k, int num_cu, int min_rounds)
{
    const bool opc = (opA == 'C' || opA == 'c' || opA == 'T' || opA == 't');
    if (!cplx) return opc ? ws_need<false, true>(m, n, k, num_cu, min_rounds) : ws_need<false, false>(m, n, k, num_cu, min_rounds);
    auto need = [&](int mm, int kk) {
        return opc ? ws_need<true, true>(mm, n, kk, num_cu, min_rounds) : ws_need<true, false>(mm, n, kk, num_cu, min_rounds);
    };
    size_t r = need(m, k);
    // a filter product with ragged rims may be cut into a 3M bulk and two thin 4M rims (launch_gemm): cover those shapes too
    const bool split = opc ? split3m_applies<true, true>(m, n, k, true) : split3m_applies<true, false>(m, n, k, true);
    if (split) {
        const int m1 = m - m % 128, k1 = k - k % 8;
        r = std::max(r, need(m1, k1));
        if (k1 < k) r = std::max(r, need(m1, k - k1));
        if (m1 < m) r = std::max(r, need(m - m1, k));
    }
    return r;
}

int gemm_f64(hipStream_t st, bool cplx, char opA, int m, int n, int k, const double* alpha, const double* A, long lda,
             const double* B, long ldb, const double* beta, double* C, long ldc, double* ws, size_t ws_bytes, int num_cu,
             int tag, int device, double* exec_flops, int min_rounds)
{
    const bool opc = (opA == 'C' || opA == 'c' || opA == 'T' || opA == 't');
    const LaunchInfo li{device, exec_flops, min_rounds};
    // tag 2 = the H-times-block products of Rayleigh-Ritz and of the residual step: three multiplications like the filter
    // since round 4 (the residuals that decide locking are re-taken from a fresh four-product H v whenever they come within
    // 1e-3 of the tolerance, chase_hip_impl.hpp Resd; CHASE_HIP_GEMM3M_RR=0 restores four products); tag 3 = verification
    // products (recompute_residuals, the borderline re-check): four multiplications, always
    static const bool rr3m = [] { const char* e = getenv("CHASE_HIP_GEMM3M_RR"); return e ? atoi(e) != 0 : true; }();
    const bool allow2 = (tag == 2) && rr3m;
#define CHASE_GEMM_DISPATCH(CP, OC)                                                                                    \
    (tag == 1 ? launch_gemm<CP, OC, 1>(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, ws, ws_bytes, num_cu, true, li) \
              : launch_gemm<CP, OC, 0>(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, ws, ws_bytes, num_cu, allow2, li))
    if (!cplx) return opc ? CHASE_GEMM_DISPATCH(false, true) : CHASE_GEMM_DISPATCH(false, false);
    return opc ? CHASE_GEMM_DISPATCH(true, true) : CHASE_GEMM_DISPATCH(true, false);
#undef CHASE_GEMM_DISPATCH
}

// ---- register-resident MFMA peak probe ("to be confirmed by a register-resident MFMA micro-benchmark")
__global__ __launch_bounds__(256, 2) void mfma_f64_peak_kernel(double* out, int iters)
{
    // 16 independent accumulators per wave (the GEMM's count), two workgroups per CU = two waves per SIMD
    d4_t acc[16];
    #pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = d4_t{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    #pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int mfma_f64_peak(hipStream_t st, double* out, int blocks, int iters)
{
    hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, st, out, iters);
    return (int)hipGetLastError();
}

} // namespace chase_hip
