// VAE-decoder kernels that are not GEMM-shaped: GroupNorm(+SiLU) over channels-last blocks and the
// 1 -> Cout input convolution.  (The 3x3x3 convolutions, the k2s2 transposed convolution, the 1x1
// shortcut and the mid-block attention projections run on the MFMA GEMM in gemm.hip.)
#include "common.h"

// ---------------------------------------------------------------------------------------------
// GroupNorm (affine, fp32 statistics, two-pass variance) + optional SiLU on one primitive's
// [V, C] channels-last block.  256 threads: thread = (voxel part, channel); C divides 256.
template <int DT>
__global__ __launch_bounds__(256) void groupnorm_silu_kernel(const typename T16<DT>::S* __restrict__ in,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            typename T16<DT>::S* __restrict__ out, int V, int C,
                                                            int cpg, float eps, int silu) {
    using S = typename T16<DT>::S;
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int c = tid % C, part = tid / C, parts = 256 / C;
    const S* src = in + (int64_t)blockIdx.x * V * C;
    S* dst = out + (int64_t)blockIdx.x * V * C;
    const int g0 = (c / cpg) * cpg;
    const float inv_cnt = 1.0f / (float)(cpg * V);

    float s = 0.f;
    for (int v = part; v < V; v += parts) s += (float)src[(int64_t)v * C + c];
    red[tid] = s;
    __syncthreads();
    float tot = 0.f;
    for (int pp = 0; pp < parts; ++pp)
        for (int j = 0; j < cpg; ++j) tot += red[pp * C + g0 + j];
    const float mean = tot * inv_cnt;
    __syncthreads();

    float q = 0.f;
    for (int v = part; v < V; v += parts) {
        const float d = (float)src[(int64_t)v * C + c] - mean;
        q += d * d;
    }
    red[tid] = q;
    __syncthreads();
    float qt = 0.f;
    for (int pp = 0; pp < parts; ++pp)
        for (int j = 0; j < cpg; ++j) qt += red[pp * C + g0 + j];
    const float rstd = 1.0f / sqrtf(qt * inv_cnt + eps);

    const float ga = gamma[c] * rstd, be = beta[c] - mean * gamma[c] * rstd;
    for (int v = part; v < V; v += parts) {
        float y = (float)src[(int64_t)v * C + c] * ga + be;
        if (silu) y = silu_f(y);
        dst[(int64_t)v * C + c] = (S)y;
    }
}

// ---------------------------------------------------------------------------------------------
// Single-read GroupNorm(+SiLU) for the decoder's own shapes: the primitive's whole [V, C] block (32 KB or 256 KB) is
// loaded ONCE with 16-byte loads and stays in registers (NCH chunks of 8 halves per thread) through the two-pass
// statistics, the affine + SiLU and the 16-byte stores - the generic kernel above re-reads it three times with 2-byte
// accesses (profiles/r2_decode_kernel_trace_before.txt: 0.65 - 0.92 ms per call against 0.03 - 0.2 ms of HBM time).
// Chunk q = tid + NT * i covers elements 8q .. 8q+7 of the block; NT is a multiple of C/8, so a thread's 8 channels are
// the same in every chunk: channels c0 .. c0+7, c0 = (tid % (C/8)) * 8.
//   CPG8 = true : C = 256, 8 channels per group -> a chunk IS one group (group tid % 32); lanes l and l+32 share it.
//   CPG8 = false: C = 32, one channel per group -> 8 statistics per thread; lanes with equal l % 4 share them.
template <int DT, int NT, int NCH, bool CPG8>
__global__ __launch_bounds__(NT) void groupnorm_silu_reg_kernel(const typename T16<DT>::S* __restrict__ in,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                typename T16<DT>::S* __restrict__ out, float inv_cnt,
                                                                float eps, int silu) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    constexpr int NWAVE = NT / 64, NST = CPG8 ? 1 : 8, CPT = CPG8 ? 32 : 4;   // statistics per thread; distinct threads' channel sets
    __shared__ float part[NWAVE][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = tid % CPT, c0 = slot * 8;
    const int64_t base = (int64_t)blockIdx.x * NT * NCH * 8;
    V8 x[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) x[i] = *reinterpret_cast<const V8*>(in + base + (int64_t)(tid + NT * i) * 8);

    // wave reduction of NST per-thread values over the lanes sharing a channel set, then over the waves through LDS;
    // returns the block-wide totals of THIS thread's statistics
    auto reduce = [&](float (&v)[NST]) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            if (CPG8) v[k] += __shfl_xor(v[k], 32);
            else {
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) v[k] += __shfl_xor(v[k], off);
            }
        }
        __syncthreads();                                   // previous use of `part` is over
        if (lane < CPT) {
#pragma unroll
            for (int k = 0; k < NST; ++k) part[wave][lane * NST + k] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) t += part[w][slot * NST + k];
            v[k] = t;
        }
    };

    float mean[NST], rstd[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) mean[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) mean[CPG8 ? 0 : e] += (float)x[i][e];
    reduce(mean);
#pragma unroll
    for (int k = 0; k < NST; ++k) { mean[k] *= inv_cnt; rstd[k] = 0.f; }
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (float)x[i][e] - mean[CPG8 ? 0 : e];
            rstd[CPG8 ? 0 : e] += d * d;
        }
    reduce(rstd);
    float ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float r = 1.0f / sqrtf(rstd[CPG8 ? 0 : e] * inv_cnt + eps);
        const float g = gamma[c0 + e];
        ga[e] = g * r;
        be[e] = beta[c0 + e] - mean[CPG8 ? 0 : e] * g * r;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)x[i][e] * ga[e] + be[e];
            if (silu) y = silu_f(y);
            o[e] = (S)y;
        }
        *reinterpret_cast<V8*>(out + base + (int64_t)(tid + NT * i) * 8) = o;
    }
}

// Streaming form for blocks too large for registers (256 channels x 8^3 voxels = 256 KB per primitive): one pass for
// the statistics - sums of d = x - x0 and d^2 with x0 = the group's first element (a sample of the distribution, so the
// single-pass variance (sum d^2 - (sum d)^2 / n) / n has no cancellation problem) - and one pass to apply; 16-byte
// accesses throughout.  C = 256, 8 channels per group (a chunk is one group).
template <int DT, int NT>
__global__ __launch_bounds__(NT) void groupnorm_silu_stream_kernel(const typename T16<DT>::S* __restrict__ in,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   typename T16<DT>::S* __restrict__ out, int nch,
                                                                   float inv_cnt, float eps, int silu) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    constexpr int NWAVE = NT / 64;
    __shared__ float part[2][NWAVE][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = tid & 31, c0 = g * 8;
    const int64_t base = (int64_t)blockIdx.x * NT * nch * 8;
    const float x0 = (float)in[base + c0];
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < nch; ++i) {
        const V8 x = *reinterpret_cast<const V8*>(in + base + (int64_t)(tid + NT * i) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (float)x[e] - x0;
            s1 += d;
            s2 += d * d;
        }
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lane < 32) { part[0][wave][lane] = s1; part[1][wave][lane] = s2; }
    __syncthreads();
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) { t1 += part[0][w][g]; t2 += part[1][w][g]; }
    const float md = t1 * inv_cnt;                                   // mean - x0
    const float var = fmaxf(t2 * inv_cnt - md * md, 0.f);
    const float mean = x0 + md, r = 1.0f / sqrtf(var + eps);
    float ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float gm = gamma[c0 + e];
        ga[e] = gm * r;
        be[e] = beta[c0 + e] - mean * gm * r;
    }
    for (int i = 0; i < nch; ++i) {
        const int64_t off = base + (int64_t)(tid + NT * i) * 8;
        const V8 x = *reinterpret_cast<const V8*>(in + off);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)x[e] * ga[e] + be[e];
            if (silu) y = silu_f(y);
            o[e] = (S)y;
        }
        *reinterpret_cast<V8*>(out + off) = o;
    }
}

extern "C" int primx_groupnorm_silu(const void* in, const float* gamma, const float* beta, void* out, int P, int V,
                                    int C, int groups, float eps, int silu, int dtype, void* stream) {
    PRIMX_REQUIRE(in && gamma && beta && out, "primx_groupnorm_silu: null pointer");
    PRIMX_REQUIRE(P > 0 && V > 0 && C > 0 && C <= 256 && 256 % C == 0 && groups > 0 && C % groups == 0,
                  "primx_groupnorm_silu: need C | 256 and groups | C (C=%d groups=%d)", C, groups);
    const int cpg = C / groups;
    const float inv_cnt = 1.0f / (float)(cpg * V);
    hipStream_t st = (hipStream_t)stream;
#define GN_REG(NT_, NCH_, CPG8_)                                                                                        \
    PRIMX_DISPATCH_16(dtype, "primx_groupnorm_silu",                                                                    \
                      hipLaunchKernelGGL((groupnorm_silu_reg_kernel<DT, NT_, NCH_, CPG8_>), dim3(P), dim3(NT_), 0, st,  \
                                         (const typename T16<DT>::S*)in, gamma, beta, (typename T16<DT>::S*)out, inv_cnt, \
                                         eps, silu))
    if (C == 256 && cpg == 8 && V == 64) GN_REG(256, 8, true);            // 4^3 stages
    else if (C == 256 && cpg == 8 && (V * 32) % 1024 == 0)                // after the upsample: 256 KB per primitive, streamed
        PRIMX_DISPATCH_16(dtype, "primx_groupnorm_silu",
                          hipLaunchKernelGGL((groupnorm_silu_stream_kernel<DT, 1024>), dim3(P), dim3(1024), 0, st,
                                             (const typename T16<DT>::S*)in, gamma, beta, (typename T16<DT>::S*)out,
                                             V * 32 / 1024, inv_cnt, eps, silu));
    else if (C == 32 && cpg == 1 && V == 512) GN_REG(256, 8, false);      // 8^3 x 32-channel stages
    else
        PRIMX_DISPATCH_16(dtype, "primx_groupnorm_silu",
                          hipLaunchKernelGGL((groupnorm_silu_kernel<DT>), dim3(P), dim3(256), 0, st,
                                             (const typename T16<DT>::S*)in, gamma, beta, (typename T16<DT>::S*)out, V, C,
                                             cpg, eps, silu));
#undef GN_REG
    PRIMX_CHECK_LAUNCH("primx_groupnorm_silu");
    return PRIMX_OK;
}

// ---------------------------------------------------------------------------------------------
// conv_in: Conv3d(1 -> Cout, k3, p1) applied to z' = a*z + b (post_quant_conv, a 1x1x1 conv on one
// channel) with ZERO padding of z' (the padding is applied after the affine).  fp32 math, 27 taps.
// One workgroup per primitive: the zero-padded (S+2)^3 latent sits in LDS, thread = output channel with its 27
// weights in registers, the voxel loop reads LDS broadcasts and writes 2*Cout contiguous bytes per voxel
// (HBM-write-bound: 67 MB for 2048 primitives; the first version - one thread per output element, 27 guarded global
// loads each - took 850 us = 12x the write time, profiles/r2_decode_kernel_trace_before.txt).
template <int DT>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ z, float a, float b,
                                                      const float* __restrict__ W, const float* __restrict__ bias,
                                                      typename T16<DT>::S* __restrict__ out, int S, int Cout) {
    using St = typename T16<DT>::S;
    extern __shared__ __attribute__((aligned(16))) float zp[];    // (S+2)^3, zero border
    const int V = S * S * S, SP = S + 2, VP = SP * SP * SP;
    const int64_t prim = blockIdx.x;
    for (int i = threadIdx.x; i < VP; i += blockDim.x) {
        const int zz = i / (SP * SP) - 1, yy = (i / SP) % SP - 1, xx = i % SP - 1;
        const bool in = (unsigned)zz < (unsigned)S && (unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S;
        zp[i] = in ? a * z[prim * V + (zz * S + yy) * S + xx] + b : 0.f;
    }
    __syncthreads();
    for (int co = threadIdx.x; co < Cout; co += blockDim.x) {
        float w[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) w[t] = W[co * 27 + t];
        const float bv = bias[co];
        St* dst = out + prim * V * Cout + co;
        for (int v = 0; v < V; ++v) {
            const int zc = v / (S * S), yc = (v / S) % S, xc = v % S;
            const float* base = zp + (zc * SP + yc) * SP + xc;     // tap (0,0,0) of the padded block
            float acc = bv;
#pragma unroll
            for (int t = 0; t < 27; ++t) acc = fmaf(w[t], base[((t / 9) * SP + (t / 3) % 3) * SP + t % 3], acc);
            dst[(int64_t)v * Cout] = (St)acc;
        }
    }
}

// S = 4 (the shipped latent grid): no LDS at all.  A primitive's 64 latent values are wave-uniform, so they sit in registers
// (scalar loads), the 64 x 27 tap loop is unrolled at compile time and the taps that fall into the zero padding are simply
// not emitted (1000 of 1728 FMAs remain; fmaf(w, 0, acc) == acc, so the result is bit-identical to the padded loop above).
// The LDS version spent its time ISSUING broadcast reads: 27 ds_read_b32 per output voxel and thread = 7k LDS
// instructions per workgroup (94 us for 2048 primitives, against 67 MB of output).
template <int DT>
__global__ __launch_bounds__(256) void conv_in_s4_kernel(const float* __restrict__ z, float a, float b,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         typename T16<DT>::S* __restrict__ out, int Cout) {
    using St = typename T16<DT>::S;
    const int64_t prim = blockIdx.x;
    const int co = blockIdx.y * 256 + threadIdx.x;
    if (co >= Cout) return;
    float zv[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) zv[i] = a * z[prim * 64 + i] + b;
    float w[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) w[t] = W[co * 27 + t];
    const float bv = bias[co];
    St* dst = out + prim * 64 * Cout + co;
#pragma clang loop unroll(full)
    for (int v = 0; v < 64; ++v) {
        const int zc = v >> 4, yc = (v >> 2) & 3, xc = v & 3;
        float acc = bv;
#pragma clang loop unroll(full)
        for (int t = 0; t < 27; ++t) {
            const int zz = zc + t / 9 - 1, yy = yc + (t / 3) % 3 - 1, xx = xc + t % 3 - 1;
            if (zz >= 0 && zz < 4 && yy >= 0 && yy < 4 && xx >= 0 && xx < 4) acc = fmaf(w[t], zv[(zz * 4 + yy) * 4 + xx], acc);
        }
        dst[(int64_t)v * Cout] = (St)acc;
    }
}

extern "C" int primx_conv_in(const float* in, float pq_scale, float pq_bias, const float* W, const float* bias,
                             void* out, int P, int S, int Cout, int dtype, void* stream) {
    PRIMX_REQUIRE(in && W && bias && out && P > 0 && S > 0 && S <= 30 && Cout > 0, "primx_conv_in: bad argument");
    if (S == 4) {
        PRIMX_DISPATCH_16(dtype, "primx_conv_in",
                          hipLaunchKernelGGL((conv_in_s4_kernel<DT>), dim3(P, (Cout + 255) / 256), dim3(256), 0, (hipStream_t)stream, in,
                                             pq_scale, pq_bias, W, bias, (typename T16<DT>::S*)out, Cout));
        PRIMX_CHECK_LAUNCH("primx_conv_in");
        return PRIMX_OK;
    }
    const size_t lds = (size_t)(S + 2) * (S + 2) * (S + 2) * sizeof(float);
    PRIMX_DISPATCH_16(dtype, "primx_conv_in",
                      hipLaunchKernelGGL((conv_in_kernel<DT>), dim3(P), dim3(256), lds, (hipStream_t)stream, in,
                                         pq_scale, pq_bias, W, bias, (typename T16<DT>::S*)out, S, Cout));
    PRIMX_CHECK_LAUNCH("primx_conv_in");
    return PRIMX_OK;
}
