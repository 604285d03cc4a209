// Probe (round 6): what bounds the k-loop of the 256 x 288 tile at T = 32768?  The 8-wave kernel's operand stream WITHOUT its MFMAs
// and LDS reads (the "no stores" probe build of round 6 turned out to have no MFMAs either - dead accumulators - and its loop still took
// 1650 cycles per 32-wide slice: the loop is bound by the operand stream, not by the matrix pipe).  Variables:
//   KS      halves of k per LDS-DMA row segment: 32 = 64-byte requests (the shipped 4-stage ring of 32-wide slices), 64 = 128-byte
//           requests = whole cache lines (2-stage ring of 64-wide tiles, the same 139 KB)
//   NST     ring stages
//   TOUCH   0 / 1: two extra waves request one dword of every row segment the ring will want D slices later (L2 touch-ahead from
//           waves whose vmcnt nobody waits for)
//   burn    s_sleep units per slice (64 cycles each) standing in for the MFMA time (18 = 1152 cycles per 32-wide slice)
// Same tile walk as the library (xcd_tile2d, gm = 8, sub-blocks of 8 x 4 tiles), M x 4608 x 1152, rotating A buffers.
// Build: hipcc --offload-arch=gfx950 -O3 -o big_ingest big_ingest.hip ; run: ./big_ingest [M=32768]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void LV;
typedef __attribute__((address_space(1))) const void GV;

__device__ unsigned long long g_cyc[8192];
__device__ unsigned int g_sink;

__device__ __forceinline__ void xcd_tile2d(int bid, int mt, int nt, int gm, int sr, int sc, int& mi, int& ni) {
    const int gn = 8 / gm, bm = mt / gm, bn = nt / gn;
    const int x = bid & 7, local = bid >> 3;
    const int per_strip = bm * sc;
    const int st = local / per_strip, rem = local - st * per_strip;
    const int w = min(sc, bn - st * sc);
    const int g = rem / (sr * w), rem2 = rem - g * (sr * w);
    const int c = rem2 / sr, r = rem2 - c * sr;
    mi = (x / gn) * bm + g * sr + r;
    ni = (x - (x / gn) * gn) * bn + st * sc + c;
}

template <int KS, int NST, int TOUCH>
__global__ __launch_bounds__(640) void ingest(const char* __restrict__ A, const char* __restrict__ W, int M, int N, int K, int burn, int D,
                                              int sr, int sc) {
    constexpr int BM = 256, BN = 288, ROWS = BM + BN;
    constexpr int RB = KS * 2;                   // bytes per row segment
    constexpr int STAGE = ROWS * RB;
    constexpr int RPI = 1024 / RB;               // rows per 1 KiB instruction
    constexpr int NINST = ROWS / RPI;            // 34 (KS = 32) / 68 (KS = 64)
    constexpr int NSLOT = (NINST + 7) / 8;       // 5 / 9
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + 1024];   // (+ a strip the touch-ahead writes and nobody reads)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = N / BN, mt = M / BM;
    int mi_t, ni_t;
    xcd_tile2d(blockIdx.x, mt, nt, 8, sr, sc, mi_t, ni_t);
    const int m0 = mi_t * BM, n0 = ni_t * BN;
    const int nks = K / KS;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 8) {
        const char* gp[NSLOT];
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int t = min(wave + 8 * i, NINST - 1);
            const int row = RPI * t + lane / (RB / 16);
            const int c = lane % (RB / 16);
            gp[i] = (row < BM) ? A + ((size_t)(m0 + row) * K) * 2 + c * 16 : W + ((size_t)(n0 + row - BM) * K) * 2 + c * 16;
        }
        const bool last_slot = wave + 8 * (NSLOT - 1) < NINST;
        auto issue_one = [&](int ks, int stage, int i) {
            if (i < NSLOT - 1 || last_slot)
                __builtin_amdgcn_global_load_lds((GV*)(uintptr_t)(gp[i] + (size_t)ks * RB), (LV*)(smem + stage * STAGE + (wave + 8 * i) * 1024), 16, 0, 0);
        };
        constexpr int NPRE = NST == 2 ? 2 : NST - 1;
#pragma unroll
        for (int pre = 0; pre < NPRE; ++pre)
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) issue_one(min(pre, nks - 1), pre, i);
        int st = 0;
        for (int ks = 0; ks < nks; ++ks) {
            if (NST == 2) {
                // two 64-wide tiles: tile ks has landed (<= 8 outstanding = only tile ks + 1's), "compute", then refill its stage with tile ks + 2
                asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                for (int b = 0; b < burn; ++b) __builtin_amdgcn_s_sleep(1);
                asm volatile("s_barrier" ::: "memory");
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) issue_one(min(ks + 2, nks - 1), st, i);
                st ^= 1;
            } else {
                // the shipped protocol: <= 4 outstanding = slices ks and ks + 1 have landed; slice ks + 3 goes into slice ks - 1's stage
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
                const int st_fill = (st == 0) ? NST - 1 : st - 1;
                const int ks_fill = min(ks + NST - 1, nks - 1);
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) issue_one(ks_fill, st_fill, i);
                for (int b = 0; b < burn; ++b) __builtin_amdgcn_s_sleep(1);
                st = (st == NST - 1) ? 0 : st + 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (TOUCH) {
        // waves 8, 9: one dword per row segment of slice ks + D (544 segments = 8.5 wave-instructions per slice, 4 - 5 each)
        unsigned sink = 0;
        const int tw = wave - 8;
        for (int ks = 0; ks < nks; ++ks) {
            asm volatile("s_barrier" ::: "memory");
            if (NST == 2) asm volatile("s_barrier" ::: "memory");
            const int kt = ks + D;
            if (kt < nks) {
                for (int r = tw * 64 + lane; r < ROWS; r += 128) {
                    const char* p = (r < BM) ? A + ((size_t)(m0 + r) * K) * 2 : W + ((size_t)(n0 + r - BM) * K) * 2;
                    __builtin_amdgcn_global_load_lds((GV*)(uintptr_t)(p + (size_t)kt * RB), (LV*)(smem + NST * STAGE + tw * 256), 4, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (sink == 0x12345u) g_sink = sink;
    } else {
        for (int ks = 0; ks < nks; ++ks) {
            asm volatile("s_barrier" ::: "memory");
            if (NST == 2) asm volatile("s_barrier" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.x < 8192) g_cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static char* As[4];
static char* Wd;

template <int KS, int NST, int TOUCH>
static void run(const char* name, int M, int burn, int D) {
    const int N = 4608, K = 1152;
    const int mt = M / 256, nt = N / 288, G = mt * nt;
    const int bm = mt / 8;
    int sr = bm;
    if (bm > 8) for (int d = 8; d >= 1; --d) if (bm % d == 0) { sr = d; break; }
    const int sc = std::max(1, std::min(nt, 32 / sr));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int threads = 640;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((ingest<KS, NST, TOUCH>), dim3(G), dim3(threads), 0, 0, As[i & 3], Wd, M, N, K, burn, D, sr, sc);
    CK(hipDeviceSynchronize());
    const int reps = 8;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ingest<KS, NST, TOUCH>), dim3(G), dim3(threads), 0, 0, As[i & 3], Wd, M, N, K, burn, D, sr, sc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int n = std::min(G, 8192);
    std::vector<unsigned long long> c(n);
    CK(hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_cyc), n * sizeof(unsigned long long)));
    double mean = 0;
    for (auto v : c) mean += (double)v;
    mean /= n;
    const double slices32 = K / 32.0;           // normalise to 32-wide slices (34.8 KB each)
    printf("%-58s M=%6d burn=%2d D=%2d: %7.1f us per launch | %6.0f cycles per 32-wide slice (34.8 KB) = %5.1f B/clk/CU\n", name, M, burn, D,
           ms * 1e3 / reps, mean / slices32, 34816.0 / (mean / slices32));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768;
    const size_t a_bytes = (size_t)M * 1152 * 2, w_bytes = (size_t)4608 * 1152 * 2;
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&As[i], a_bytes)); CK(hipMemset(As[i], 1 + i, a_bytes)); }
    CK(hipMalloc(&Wd, w_bytes));
    CK(hipMemset(Wd, 7, w_bytes));
    for (int burn : {0, 18}) {
        run<32, 4, 0>("KS=32 (64 B rows), 4 stages [shipped ring]", M, burn, 0);
        run<64, 2, 0>("KS=64 (128 B rows), 2 stages", M, burn * 2, 0);
        run<32, 4, 1>("KS=32, 4 stages + touch-ahead waves", M, burn, 6);
        run<32, 4, 1>("KS=32, 4 stages + touch-ahead waves", M, burn, 12);
        run<32, 4, 1>("KS=32, 4 stages + touch-ahead waves", M, burn, 24);
        run<64, 2, 1>("KS=64, 2 stages + touch-ahead waves", M, burn * 2, 4);
        run<64, 2, 1>("KS=64, 2 stages + touch-ahead waves", M, burn * 2, 8);
    }
    return 0;
}
