// Probe (round 6): how long does a CU stay empty between two workgroups of a multi-round launch?  The per-round timeline of the 256 x 288
// GEMMs at T = 32768 (profiles/r6_largeM_diagnosis.txt section 6) shows 8 - 16 us between the mean end of a round of 256 workgroups and the mean
// start of the next.  Here: workgroups of 512 threads that do nothing but spin for `busy` us on s_memrealtime (optionally ending with a store
// burst of `kb` KB per workgroup), with 139 KB or 64 KB of LDS (one or two workgroups per CU), 2 x 256 .. 8 x 256 workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -o wg_turnaround wg_turnaround.hip ; run: ./wg_turnaround
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__device__ unsigned long long g_t[8192][2];

template <int LDS_KB>
__global__ __launch_bounds__(512, 2) void spin(int busy_ticks, float* sink, int kb) {
    __shared__ char lds[LDS_KB * 1024];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) lds[blockIdx.x & 1023] = 1;
    while (__builtin_amdgcn_s_memrealtime() < t0 + busy_ticks) __builtin_amdgcn_s_sleep(8);
    if (kb > 0) {   // a write burst of kb KB per workgroup (16 bytes per lane and instruction), not waited for
        float4* dst = reinterpret_cast<float4*>(sink) + (size_t)blockIdx.x * (kb * 64) + threadIdx.x;
        for (int i = 0; i < kb * 64 / 512; ++i) dst[i * 512] = float4{1.f, 2.f, 3.f, (float)lds[threadIdx.x & 1023]};
    }
    if (threadIdx.x == 0) { g_t[blockIdx.x][0] = t0; g_t[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime(); }
}

template <int LDS_KB>
static void run(int wgs, int busy_us, int kb, float* sink) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((spin<LDS_KB>), dim3(wgs), dim3(512), 0, 0, busy_us * 100, sink, kb);
        hipDeviceSynchronize();
    }
    static unsigned long long t[8192][2];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(unsigned long long) * 2 * wgs);
    std::vector<int> ord(wgs);
    for (int i = 0; i < wgs; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return t[a][0] < t[b][0]; });
    const unsigned long long z = t[ord[0]][0];
    unsigned long long last = 0;
    for (int i = 0; i < wgs; ++i) last = std::max(last, t[i][1]);
    printf("LDS %3d KB, %4d workgroups x %3d us busy, %4d KB stored each: first start -> last end %7.1f us; rounds of 256 in start order, mean start / mean end:",
           LDS_KB, wgs, busy_us, kb, (last - z) * 0.01);
    for (int r0 = 0; r0 < wgs; r0 += 256) {
        double s = 0, e = 0;
        const int n = std::min(256, wgs - r0);
        for (int i = r0; i < r0 + n; ++i) { s += (t[ord[i]][0] - z) * 0.01; e += (t[ord[i]][1] - z) * 0.01; }
        printf("  %.1f / %.1f", s / n, e / n);
    }
    printf("\n");
}

int main() {
    float* sink;
    hipMalloc(&sink, (size_t)2048 * 1024 * 1024);
    for (int kb : {0, 144, 576}) {
        run<139>(512, 40, kb, sink);
        run<139>(2048, 40, kb, sink);
        run<64>(512, 40, kb, sink);
        run<64>(2048, 40, kb, sink);
    }
    run<139>(512, 150, 576, sink);
    return 0;
}
