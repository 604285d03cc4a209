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

extern "C" int primx_groupnorm_silu(const void* in, const float* gamma, const float* beta, void* out, int P, int V,
                                    int C, int groups, float eps, int silu, int dtype, void* stream) {
    PRIMX_REQUIRE(in && gamma && beta && out, "primx_groupnorm_silu: null pointer");
    PRIMX_REQUIRE(P > 0 && V > 0 && C > 0 && C <= 256 && 256 % C == 0 && groups > 0 && C % groups == 0,
                  "primx_groupnorm_silu: need C | 256 and groups | C (C=%d groups=%d)", C, groups);
    PRIMX_DISPATCH_16(dtype, "primx_groupnorm_silu",
                      hipLaunchKernelGGL((groupnorm_silu_kernel<DT>), dim3(P), dim3(256), 0, (hipStream_t)stream,
                                         (const typename T16<DT>::S*)in, gamma, beta, (typename T16<DT>::S*)out, V, C,
                                         C / groups, eps, silu));
    PRIMX_CHECK_LAUNCH("primx_groupnorm_silu");
    return PRIMX_OK;
}

// ---------------------------------------------------------------------------------------------
// conv_in: Conv3d(1 -> Cout, k3, p1) applied to z' = a*z + b (post_quant_conv, a 1x1x1 conv on one
// channel) with ZERO padding of z' (the padding is applied after the affine).  fp32 math, 27 taps.
template <int DT>
__global__ void conv_in_kernel(const float* __restrict__ z, float a, float b, const float* __restrict__ W,
                               const float* __restrict__ bias, typename T16<DT>::S* __restrict__ out, int64_t total,
                               int S, int Cout) {
    using St = typename T16<DT>::S;
    const int V = S * S * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const int64_t pv = i / Cout;
        const int v = (int)(pv % V);
        const int64_t p = pv / V;
        const int zc = v / (S * S), yc = (v / S) % S, xc = v % S;
        float acc = bias[co];
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int zz = zc + tap / 9 - 1, yy = yc + (tap / 3) % 3 - 1, xx = xc + tap % 3 - 1;
            if ((unsigned)zz < (unsigned)S && (unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S) {
                const float val = a * z[p * V + (zz * S + yy) * S + xx] + b;
                acc = fmaf(W[co * 27 + tap], val, acc);
            }
        }
        out[i] = (St)acc;
    }
}

extern "C" int primx_conv_in(const float* in, float pq_scale, float pq_bias, const float* W, const float* bias,
                             void* out, int P, int S, int Cout, int dtype, void* stream) {
    PRIMX_REQUIRE(in && W && bias && out && P > 0 && S > 0 && Cout > 0, "primx_conv_in: bad argument");
    const int64_t total = (int64_t)P * S * S * S * Cout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    PRIMX_DISPATCH_16(dtype, "primx_conv_in",
                      hipLaunchKernelGGL((conv_in_kernel<DT>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, in,
                                         pq_scale, pq_bias, W, bias, (typename T16<DT>::S*)out, total, S, Cout));
    PRIMX_CHECK_LAUNCH("primx_conv_in");
    return PRIMX_OK;
}
