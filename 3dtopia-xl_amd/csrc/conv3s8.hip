// 3x3x3 / stride 1 / pad 1 convolution 256 -> 32 channels on the 8^3 grid: conv1 of the first ResnetBlock of
// up_blocks[1] (models/vae3d_dib.py:62-75 inside Decoder.forward :262-270), the decode's
// second-largest kernel.  INPUT-STATIONARY, with the taps applied as a scatter into an LDS accumulator:
//
//   workgroup = one primitive, wave w = input z-plane w: its 64 voxels x 256 channels are 32 KB = 128 VGPRs per lane, held
//     as MFMA operand fragments a[ks][cg] (16 voxels = rows y = 2 cg, 2 cg + 1 of the plane; 32 channels) and read from
//     memory exactly once - the implicit GEMM of gemm.hip gathers every activation 27 times (N = 32 output channels cannot
//     amortise that: it ran at 380 TFLOP/s, bound by the gather);
//   per tap (dz, dy, dx): Y = W_tap x X over all 256 channels (64 MFMAs 16x16x32 into 32 accumulator registers), then
//     out[z - dz][y - dy][x - dx] += Y[y][x] as a 16-byte read-modify-write into a 64 KB fp32 image of the primitive's
//     output in LDS.  The shift is ADDRESS ARITHMETIC (no DPP, no zero padding: out-of-volume lanes are masked off), so
//     the tap loop is an ordinary dynamic loop;
//   taps are ordered by dz (0, +1, -1): within a phase exactly ONE wave adds into a given output plane (plane z gets its
//     dz-phase contribution from input plane z + dz), so the sums are formed in a fixed order - deterministic, no
//     contention - and the per-tap barrier of the weight ring separates the phases.  Boundary waves skip the taps that
//     would land outside the volume (plane 0 has no dz = +1 target, plane 7 no dz = -1): 8 % fewer MFMAs than zero padding.
//
// Weights: primx_conv3d_s8_pack turns Wk[32][6912] (k = tap * 256 + ci) into the LDS image of every (tap, 64-channel
// group) tile [32 rows][64 k] with gemm.hip's bank swizzle and the operand-row permutation that gives a lane 8 consecutive
// output channels; a tap is 16 KB of consecutive memory, copied by 16 LDS-DMA instructions (2 per wave) into a 4-tap ring.
// LDS accumulator layout: [voxel][8 chunks of 4 floats]; the chunk (ni << 2 | lg) holds channels lg * 8 + ni * 4 .. + 4 and
// is stored at chunk position (ni << 2 | lg) XOR ((voxel >> 1) & 7), which spreads the 16 lanes of a ds_read_b128 /
// ds_write_b128 lane group (consecutive voxels, lg in {0,1} or {2,3}) over the 16 sixteen-byte bank slots.
#include <stdlib.h>

#include "common.h"

namespace {

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// Arguments of the FUSED form (conv1 of up_blocks[1].nets[0] with everything around it, vae3d_dib.py:109-125): `in` is the RAW
// upsample output; its GroupNorm(32 groups of 8 channels) statistics arrive as the 16 partial shifted sums per (primitive,
// group) that convt.hip wrote (shift = up_bias[8 g]); the kernel normalises + SiLUs its plane in registers after it has
// used the raw plane for the 1x1 shortcut convolution (weight block 27 of the image, output `sc_out`).
template <typename S>
struct FusedArgs {
    const float* part;       // [P][16][32][2]
    const S* up_bias;        // [256] bias of the producing upsample (the statistics' shifts)
    const float* gamma;      // [256]
    const float* beta;       // [256]
    float eps;
    const S* sc_bias;        // [32] or null
    S* sc_out;               // [P][512][32]
};

__device__ __forceinline__ float silu_fast(float v) {   // v * sigmoid(v) with v_exp_f32 + v_rcp_f32 (1 ulp) - silu_f's IEEE
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));   // division is ~12 VALU instructions
}

template <int DT, int FUSED>
__global__ __launch_bounds__(512) void conv3_s8c256n32_kernel(const typename T16<DT>::S* __restrict__ in,
                                                             const typename T16<DT>::S* __restrict__ Wp,
                                                             const typename T16<DT>::S* __restrict__ bias,
                                                             const typename T16<DT>::S* __restrict__ res, float res_scale,
                                                             typename T16<DT>::S* __restrict__ out,
                                                             const FusedArgs<typename T16<DT>::S> fa) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    typedef __attribute__((address_space(3))) void LV;
    constexpr int CIN = 256, COUT = 32, VOX = 512, NST = 4, TAPB = 4 * 32 * 64;   // halves per tap block (16 KB)
    __shared__ __attribute__((aligned(16))) float obuf[VOX * COUT];               // 64 KB
    __shared__ __attribute__((aligned(16))) S wring[NST * TAPB];                  // 64 KB
    __shared__ __attribute__((aligned(16))) float gab[FUSED ? 2 * 256 : 4];       // FUSED: per-channel scale / shift of the normalisation

    const int tid = threadIdx.x, lane = tid & 63;
    const int zi = __builtin_amdgcn_readfirstlane(tid >> 6);                      // wave = input plane
    const int j = lane & 15, lg = lane >> 4;
    const int prim = blockIdx.x;

    // ---- zero the accumulator image (one voxel row per thread)
    {
        f32x4* o4 = reinterpret_cast<f32x4*>(obuf) + tid * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) o4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if constexpr (FUSED) {
        // per-channel scale and shift from the 16 partial sums of the channel's group, added in index order
        if (tid < 256) {
            const int g = tid >> 3;
            const float* pp = fa.part + ((int64_t)prim * 16 * 32 + g) * 2;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s1 += pp[i * 64]; s2 += pp[i * 64 + 1]; }
            const float m = s1 * (1.0f / 4096.0f), var = s2 * (1.0f / 4096.0f) - m * m;
            const float mean = (float)fa.up_bias[8 * g] + m;
            const float r = 1.0f / sqrtf(fmaxf(var, 0.f) + fa.eps);
            const float ga = fa.gamma[tid] * r;
            gab[tid] = ga;
            gab[256 + tid] = fa.beta[tid] - mean * ga;
        }
    }

    // ---- weight ring: step t handles tap (dzs + 1) * 9 + t % 9 with dzs = 0, +1, -1 for t / 9 = 0, 1, 2
    // FUSED: one more step in front - weight block 27, the shortcut - so taps are steps 1 .. 27
    constexpr int NSTEP = 27 + FUSED;
    auto tap_of = [](int t) { return (t < 9 ? 9 : t < 18 ? 18 : 0) + (t < 9 ? t : t < 18 ? t - 9 : t - 18); };
    auto block_of = [&](int s) { return FUSED ? (s == 0 ? 27 : tap_of(s - 1)) : tap_of(s); };
    const unsigned lds_w0 = (unsigned)(uintptr_t)(LV*)wring + (unsigned)zi * 2048u;
    const unsigned voff = (unsigned)(zi * 2048 + lane * 16);
    auto issue = [&](int t, int stage) {   // (SGPR base + 32-bit lane offset; M0 = LDS byte address of the 1 KB piece)
        const char* sb = reinterpret_cast<const char*>(Wp) + (int64_t)__builtin_amdgcn_readfirstlane(block_of(t)) * (TAPB * 2);
        const unsigned m0a = __builtin_amdgcn_readfirstlane(lds_w0 + (unsigned)stage * (TAPB * 2));
        // (no instruction offset on the second piece: for LDS-DMA that field is added to the LDS address as well)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                     ::"s"(m0a), "v"(voff), "v"(voff + 1024u), "s"(sb) : "memory");
    };
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    // ---- this wave's input plane: 32 fragments of 16 voxels x 32 channels
    V8 a[8][4];
    {
        const S* src = in + ((int64_t)prim * VOX + zi * 64 + j) * CIN + lg * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) a[ks][cg] = *reinterpret_cast<const V8*>(src + cg * 16 * CIN + ks * 32);
        // All of them are made "used" HERE, so hipcc waits for them here, once: left to their first uses inside the tap loop
        // (which has a skip path, so "maybe still pending" survives every iteration) it repeated the whole countdown
        // vmcnt(31) .. vmcnt(0) in every tap, and vmcnt being one in-order counter, each tap then also drained the weight
        // DMA issued a few instructions earlier - no prefetch left (4.5 ms instead of 0.3).
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) asm volatile("" ::"v"(a[ks][cg]));
    }

    const int w_row = j;                 // operand row of the weight fragment
    const int yb = j >> 3, x = j & 7;    // in-plane position of this lane's voxel: y = 2 cg + yb
    const int lg_sw = ((lg >> 1) << 1) | (lg & 1);   // (= lg: chunk index bits of this lane's channel group, see the layout note)

    auto tap_step = [&](int t) {
        const int stage = t & 3;
        // taps <= t landed for this wave (t + 1, t + 2 may stay in flight: 2 DMA instructions each); this wave's adds of the
        // previous tap are done; after the barrier the DMA below may overwrite the stage of tap t - 1
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue(min(t + 3, NSTEP - 1), (t + 3) & 3);                       // (past the end: a harmless reload of the last block)
        if constexpr (FUSED) {
            if (t == 0) return;                                          // (step 0 is the shortcut, handled before the loop)
        }
        const int tt = t - FUSED;                                        // tap sequence number
        const int ph = tt < 9 ? 0 : tt < 18 ? 1 : 2, dydx = tt - 9 * ph;
        const int dzs = ph == 0 ? 0 : ph == 1 ? 1 : -1;
        const int dy = dydx / 3 - 1, dx = dydx - 3 * (dydx / 3) - 1;
        const int zo = zi - dzs;                                          // out[z] += W[dz] * in[z + dz]
        if ((unsigned)zo >= 8u) return;                                   // wave-uniform: this plane has no target for this dz

        f32x4 acc[4][2];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) acc[cg][0] = acc[cg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const S* Ws = wring + stage * TAPB;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const S* Wt = Ws + (ks >> 1) * (32 * 64);
            const V8 w0 = *reinterpret_cast<const V8*>(Wt + lds_off(w_row, (ks & 1) * 4 + lg));
            const V8 w1 = *reinterpret_cast<const V8*>(Wt + lds_off(w_row + 16, (ks & 1) * 4 + lg));
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                acc[cg][0] = T16<DT>::mfma16(w0, a[ks][cg], acc[cg][0]);
                acc[cg][1] = T16<DT>::mfma16(w1, a[ks][cg], acc[cg][1]);
            }
        }
        // ---- scatter-add: lane (j, lg) holds voxel (y = 2 cg + yb, x) and channels lg * 8 + ni * 4 + r.  Plain read-modify-
        // write of one 16-byte chunk per (cg, ni) - this wave is the only writer of the target plane in this phase - under the
        // lane's validity (exec mask).  (ds_add_f32 measured ~166 cycles per wave-instruction: LDS float atomics are
        // serialised per lane; the kernel took 4.5 ms with them.)
        __builtin_amdgcn_sched_barrier(0);   // the read-modify-writes stay behind the tap's MFMAs
        const int xo = x - dx;
        const bool xok = (unsigned)xo < 8u;
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            const int yo = 2 * cg + yb - dy;
            if (xok && (unsigned)yo < 8u) {
                const int vo = zo * 64 + yo * 8 + xo;
                float* row = obuf + vo * 32;
                const int sw = (vo >> 1) & 7;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    f32x4* p4 = reinterpret_cast<f32x4*>(row + ((((ni << 2) | lg_sw) ^ sw) << 2));
                    *p4 = *p4 + acc[cg][ni];
                }
            }
        }
    };
    // NOT unrolled / peeled: with hipcc's default (it peels the first three taps and specialises the loop by phase) the
    // contributions of steps 1..5 were lost on the GPU (tools/probe/conv_s8_taps.py: one nonzero tap at a time) although
    // the peeled ISA reads correctly; the rolled loop is also a third of the code.
    if constexpr (FUSED) {
        // ---- step 0: 1x1 shortcut on the RAW plane (weight block 27 in stage 0), straight to memory; then the plane is
        // normalised in place: 256 values per lane, scale / shift of its 8 channels per k-step from LDS
        tap_step(0);                                                     // wait + barrier + DMA of step 3 (returns before any tap work)
        f32x4 sacc[4][2];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) sacc[cg][0] = sacc[cg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const S* Wt = wring + (ks >> 1) * (32 * 64);
            const V8 w0 = *reinterpret_cast<const V8*>(Wt + lds_off(w_row, (ks & 1) * 4 + lg));
            const V8 w1 = *reinterpret_cast<const V8*>(Wt + lds_off(w_row + 16, (ks & 1) * 4 + lg));
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                sacc[cg][0] = T16<DT>::mfma16(w0, a[ks][cg], sacc[cg][0]);
                sacc[cg][1] = T16<DT>::mfma16(w1, a[ks][cg], sacc[cg][1]);
            }
        }
        V8 sb8 = V8{};
        if (fa.sc_bias) sb8 = *reinterpret_cast<const V8*>(fa.sc_bias + lg * 8);
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            V8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (S)(sacc[cg][e >> 2][e & 3] + (float)sb8[e]);
            *reinterpret_cast<V8*>(fa.sc_out + ((int64_t)prim * VOX + zi * 64 + cg * 16 + j) * COUT + lg * 8) = o;
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gab + ks * 32 + lg * 8), g1 = *reinterpret_cast<const f32x4*>(gab + ks * 32 + lg * 8 + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(gab + 256 + ks * 32 + lg * 8), h1 = *reinterpret_cast<const f32x4*>(gab + 256 + ks * 32 + lg * 8 + 4);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    a[ks][cg][e] = (S)silu_fast((float)a[ks][cg][e] * (e < 4 ? g0[e & 3] : g1[e & 3]) + (e < 4 ? h0[e & 3] : h1[e & 3]));
        }
    }
#pragma clang loop unroll(disable)
    for (int t = FUSED; t < NSTEP; ++t) tap_step(t);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- epilogue: one voxel per thread, 32 channels = 64 contiguous bytes of the output
    const int v = tid;
    float y[COUT];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {                     // logical chunk (ni << 2 | lg) = channels lg * 8 + ni * 4 .. + 4
        const f32x4 q = *reinterpret_cast<const f32x4*>(obuf + v * 32 + ((ch ^ ((v >> 1) & 7)) << 2));
        const int c0 = (ch & 3) * 8 + (ch >> 2) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[c0 + r] = q[r];
    }
    const int64_t off = ((int64_t)prim * VOX + v) * COUT;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        V8 bq = V8{}, rq = V8{};
        if (bias) bq = *reinterpret_cast<const V8*>(bias + 8 * q);
        if (res) rq = *reinterpret_cast<const V8*>(res + off + 8 * q);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = y[8 * q + e] + (float)bq[e];
            if (res) t += (float)rq[e];
            o[e] = (S)(t * res_scale);
        }
        *reinterpret_cast<V8*>(out + off + 8 * q) = o;
    }
}

// Wk [32][27 * 256] (k = tap * 256 + ci) -> Wp[tap][kc][rho][slot][8]; one 16-byte chunk per thread.  Block 27 (optional):
// the 1x1 shortcut weight Wsc [32][256].
__global__ __launch_bounds__(256) void conv3_s8_pack_kernel(const unsigned short* __restrict__ Wk, const unsigned short* __restrict__ Wsc,
                                                           unsigned short* __restrict__ Wp) {
    const int cid = blockIdx.x * 256 + threadIdx.x;       // chunk index in Wp
    if (cid >= (Wsc ? 28 : 27) * 4 * 32 * 8) return;
    const int slot = cid & 7, rho = (cid >> 3) & 31, kc = (cid >> 8) & 3, tap = cid >> 10;
    const int i16 = rho & 15, ni = rho >> 4;
    const int n = (i16 >> 2) * 8 + ni * 4 + (i16 & 3);
    const int c = slot ^ ((rho >> 1) & 7);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const unsigned short* src = tap < 27 ? Wk + (int64_t)n * (27 * 256) + tap * 256 + kc * 64 + c * 8 : Wsc + n * 256 + kc * 64 + c * 8;
    *reinterpret_cast<u4*>(Wp + (int64_t)cid * 8) = *reinterpret_cast<const u4*>(src);
}

}  // namespace

extern "C" int primx_conv3d_s8_pack(const void* Wk, const void* Wsc, void* Wp, int dtype, void* stream) {
    PRIMX_REQUIRE(Wk && Wp && Wk != Wp && Wsc != Wp, "primx_conv3d_s8_pack: null or aliased pointer");
    PRIMX_REQUIRE(dtype == PRIMX_F16 || dtype == PRIMX_BF16, "primx_conv3d_s8_pack: dtype must be PRIMX_F16 or PRIMX_BF16");
    hipLaunchKernelGGL(conv3_s8_pack_kernel, dim3(28 * 4 * 32 * 8 / 256), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)Wk, (const unsigned short*)Wsc, (unsigned short*)Wp);
    PRIMX_CHECK_LAUNCH("primx_conv3d_s8_pack");
    return PRIMX_OK;
}

extern "C" int primx_conv3d_s8_packed(const void* in, const void* Wp, const void* bias, const void* res, float res_scale,
                                      void* out, int P, int dtype, void* stream) {
    PRIMX_REQUIRE(in && Wp && out, "primx_conv3d_s8_packed: null pointer");
    PRIMX_REQUIRE(P > 0, "primx_conv3d_s8_packed: need P > 0 (P=%d)", P);
    PRIMX_DISPATCH_16(dtype, "primx_conv3d_s8_packed", {
        using Sx = typename T16<DT>::S;
        hipLaunchKernelGGL((conv3_s8c256n32_kernel<DT, 0>), dim3(P), dim3(512), 0, (hipStream_t)stream, (const Sx*)in, (const Sx*)Wp,
                           (const Sx*)bias, (const Sx*)res, res_scale, (Sx*)out, FusedArgs<Sx>{});
    });
    PRIMX_CHECK_LAUNCH("primx_conv3d_s8_packed");
    return PRIMX_OK;
}

extern "C" int primx_conv3d_s8_fused(const void* in_raw, const void* Wp28, const void* bias, const float* part, const void* up_bias,
                                     const float* gamma, const float* beta, float eps, const void* sc_bias, void* out, void* sc_out,
                                     int P, int dtype, void* stream) {
    PRIMX_REQUIRE(in_raw && Wp28 && part && up_bias && gamma && beta && out && sc_out, "primx_conv3d_s8_fused: null pointer");
    PRIMX_REQUIRE(P > 0, "primx_conv3d_s8_fused: need P > 0 (P=%d)", P);
    PRIMX_DISPATCH_16(dtype, "primx_conv3d_s8_fused", {
        using Sx = typename T16<DT>::S;
        FusedArgs<Sx> fa = {part, (const Sx*)up_bias, gamma, beta, eps, (const Sx*)sc_bias, (Sx*)sc_out};
        hipLaunchKernelGGL((conv3_s8c256n32_kernel<DT, 1>), dim3(P), dim3(512), 0, (hipStream_t)stream, (const Sx*)in_raw, (const Sx*)Wp28,
                           (const Sx*)bias, (const Sx*)nullptr, 1.0f, (Sx*)out, fa);
    });
    PRIMX_CHECK_LAUNCH("primx_conv3d_s8_fused");
    return PRIMX_OK;
}
