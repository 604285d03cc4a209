// Probe: how many operand bytes per clock does ONE CU take in, by delivery form, with all 256 CUs doing it at once?
// It answers the question behind "load the pre-shuffled weights global -> VGPR and keep only A on the LDS-DMA ring" (DESIGN.md
// section 4 item 12) with cycle counts instead of a byte-count argument.  No MFMAs, no LDS reads: the pure ingest ceiling of the
// k-loop of the 128 x 144 kernel (csrc/gemm.hip, gemm144l_dma_kernel: 8 compute waves + 2 loader waves, k-tiles of 64 halves =
// 128-byte row segments: A 128 rows = 16 KiB, W 144 rows = 18 KiB per k-tile).
//   mode 0  A and W by LDS-DMA from the two loader waves (34 x global_load_lds_dwordx4 of 1 KiB per k-tile) - the shipped loop's traffic
//   mode 1  A by LDS-DMA (loaders); W pre-shuffled, global_load_dwordx4 into VGPRs of the 8 compute waves, each fragment ONCE
//           (waves split N: 18 KiB per k-tile through the vector-memory path)
//   mode 2  as 1, every W fragment fetched by FOUR waves (waves split M as today: 72 KiB per k-tile)
//   mode 3  W only, as in 1 (plain-load rate)           mode 4  A only by LDS-DMA (DMA rate, half the bytes)
//   mode 5  W only by LDS-DMA from the loaders (18 KiB) mode 6  W only, as in 2 (72 KiB of plain loads)
//   mode 7  W only by LDS-DMA from the ROW-MAJOR [1152][K] matrix (8 rows x 128 B per instruction, what the shipped kernels do) -
//           against mode 5: does a pre-packed weight image (1 KiB contiguous per instruction) enter the CU faster?
//   mode 8  A + row-major W by LDS-DMA: the shipped loop's traffic exactly (mode 0 has the packed W image)
// Build: hipcc --offload-arch=gfx950 -O3 -o cu_ingest cu_ingest.hip ; run: ./cu_ingest [K=1152]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void LV;

__device__ unsigned long long g_cyc[4096];
__device__ unsigned int g_sink;

template <int MODE>
__global__ __launch_bounds__(640) void ingest(const char* __restrict__ A, const char* __restrict__ Wp, int K, int nk) {
    constexpr int STAGE = 34 * 1024, NST = 3;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 32 x 8 tiles of a 4096 x 1152 output; workgroup id -> XCD id & 7: the 32 workgroups of an XCD take 4 row tiles x all 8
    // column tiles, so an XCD's L2 holds 4 A panels + W (the real kernels' xcd_remap does the same; with the naive id -> tile map
    // every XCD fetched ALL of A and the A modes measured the fabric: 12.7 B/clk/CU)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int mt = xcd * 4 + (jx >> 3), nt = jx & 7;
    constexpr bool DMA_A = MODE == 0 || MODE == 1 || MODE == 2 || MODE == 4 || MODE == 8;
    constexpr bool DMA_W = MODE == 0 || MODE == 5 || MODE == 7 || MODE == 8;
    constexpr bool W_ROWS = MODE == 7 || MODE == 8;      // W by LDS-DMA from the row-major [1152][K] matrix (8 rows x 128 B per instruction)
    constexpr int W_REP = (MODE == 1 || MODE == 3) ? 1 : (MODE == 2 || MODE == 6) ? 4 : 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned int sink = 0;
    if (wave >= 8) {
        // ---- loader waves: instruction i of a k-tile = 1 KiB; A: 8 rows x 128 B per instruction, W: 1 KiB of the packed image
        const int lw = wave - 8;
        const unsigned lds0 = (unsigned)(uintptr_t)(LV*)smem;
        const char* a_base = A + ((size_t)mt * 128 + (lane >> 3)) * (size_t)K * 2 + (lane & 7) * 16;
        const char* w_base = W_ROWS ? Wp + ((size_t)nt * 144 + (lane >> 3)) * (size_t)K * 2 + (lane & 7) * 16
                                    : Wp + (size_t)nt * nk * 18432 + lane * 16;
        constexpr int NI = (DMA_A ? 16 : 0) + (DMA_W ? 18 : 0);
        if (NI > 0) {
            for (int t = 0; t < nk; ++t) {
                const int st = t % NST;
#pragma unroll
                for (int ii = 0; ii < NI / 2; ++ii) {
                    const int i = 2 * ii + lw;
                    const bool is_a = DMA_A && i < 16;
                    const int iw = DMA_A ? i - 16 : i;
                    const char* src = is_a ? a_base + (size_t)(i * 8) * K * 2 + (size_t)t * 128
                                      : W_ROWS ? w_base + (size_t)(iw * 8) * K * 2 + (size_t)t * 128 : w_base + ((size_t)t * 18 + iw) * 1024;
                    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + st * STAGE + i * 1024);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory");
                }
                // at most two k-tiles of this wave in flight (the ring of the real kernel holds three)
                if (NI / 2 == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
                else if (NI / 2 == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else if (W_REP > 0) {
        // ---- compute waves: W fragments of 1 KiB (16 bytes per lane, contiguous per instruction) straight into registers.
        // W_REP = 1: fragment f of the 18 goes to wave f % 8 (3, 3, 2, 2, ... per wave); W_REP = 4: the waves of one M-group
        // (wave & 3... here: wave >> 2 selects the half of the fragments, 9 each) all fetch the same 9 fragments
        const char* w_base = Wp + (size_t)nt * nk * 18432 + lane * 16;
        constexpr int NF = W_REP == 1 ? 3 : 9;
        u32x4 f[2][NF];
        auto issue = [&](int t, int b) {
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                int fr = W_REP == 1 ? wave + 8 * j : (wave >> 2) * 9 + j;
                if (fr > 17) fr = 17;                                       // (waves 2..7 re-read the last fragment: 24 instead of 18 KiB)
                f[b][j] = *reinterpret_cast<const u32x4*>(w_base + ((size_t)t * 18 + fr) * 1024);
            }
        };
        issue(0, 0);
        for (int t = 0; t < nk; t += 2) {
            issue(t + 1 < nk ? t + 1 : t, 1);
#pragma unroll
            for (int j = 0; j < NF; ++j) sink ^= f[0][j][0] ^ f[0][j][1] ^ f[0][j][2] ^ f[0][j][3];
            issue(t + 2 < nk ? t + 2 : t, 0);
#pragma unroll
            for (int j = 0; j < NF; ++j) sink ^= f[1][j][0] ^ f[1][j][1] ^ f[1][j][2] ^ f[1][j][3];
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) sink ^= f[0][j][0];
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (sink == 0x12345678u) g_sink = sink;
    if (tid == 0) g_cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
static void run(const char* name, const char* A, const char* W, int K, double bytes_per_tile) {
    const int nk = K / 64;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(ingest<MODE>, dim3(256), dim3(640), 0, 0, A, W, K, nk);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> c(256);
    CK(hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_cyc), 256 * sizeof(unsigned long long)));
    std::sort(c.begin(), c.end());
    double mean = 0;
    for (auto v : c) mean += (double)v;
    mean /= 256.0;
    printf("%-46s K=%5d: cycles per k-tile  min %7.0f  median %7.0f  mean %7.0f  max %7.0f   -> %5.1f B/clk/CU at the mean (%.0f KiB per k-tile)\n",
           name, K, (double)c[0] / nk, (double)c[128] / nk, mean / nk, (double)c[255] / nk, bytes_per_tile / (mean / nk), bytes_per_tile / 1024.0);
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 1152;
    char *A, *W;
    const size_t a_bytes = (size_t)4096 * K * 2, w_bytes = (size_t)8 * (K / 64) * 18432;
    CK(hipMalloc(&A, a_bytes));
    CK(hipMalloc(&W, w_bytes));
    CK(hipMemset(A, 1, a_bytes));
    CK(hipMemset(W, 2, w_bytes));
    run<8>("8: A + row-major W by LDS-DMA (shipped form)", A, W, K, 34816.0);
    run<0>("0: A + packed W by LDS-DMA (2 loader waves)", A, W, K, 34816.0);
    run<7>("7: W only by LDS-DMA, row-major source", A, W, K, 18432.0);
    run<4>("4: A only by LDS-DMA", A, W, K, 16384.0);
    run<5>("5: W only by LDS-DMA, packed source", A, W, K, 18432.0);
    run<3>("3: W only, global->VGPR, each fragment once", A, W, K, 24576.0);
    run<6>("6: W only, global->VGPR, each fragment x4", A, W, K, 73728.0);
    run<1>("1: A by LDS-DMA + W global->VGPR once", A, W, K, 16384.0 + 24576.0);
    run<2>("2: A by LDS-DMA + W global->VGPR x4", A, W, K, 16384.0 + 73728.0);
    return 0;
}
