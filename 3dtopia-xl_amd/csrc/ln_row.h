// LayerNorm (no affine) + adaLN modulate + cast of ONE row by ONE half-wave (32 lanes): the row body shared by
// ln_modulate_row32_kernel (rowops.hip) and the LayerNorm tail of the gate-residual GEMM (gemm.hip, EPI_GATE_RESIDUAL_LN),
// so that both produce the same bits for a row: same loads, same summation order, same rounding points.
// Replaces nn.LayerNorm(elementwise_affine=False, eps=1e-6) + modulate(): models/dit_crossattn.py:32-36,55-57,67,76,
// models/utils.py:19-20.
#pragma once
#include "common.h"

// 5-step xor reduction inside a 32-lane half of the wave
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// D = NCH4 * 128 columns; lane l32 of the half-wave owns columns 4 l32 + 128 c .. + 3 (16-byte loads, 8-byte stores).
// xr / orow: the row's first element; sh / sc: the batch entry's shift / scale vectors (16-bit, 8-byte aligned).
template <int DT, int NCH4>
__device__ __forceinline__ void ln_row32(const float* __restrict__ xr, const typename T16<DT>::S* __restrict__ sh,
                                         const typename T16<DT>::S* __restrict__ sc, typename T16<DT>::S* __restrict__ orow,
                                         int l32, float eps) {
    using S = typename T16<DT>::S;
    using V4 = typename T16<DT>::V4;
    constexpr int D = NCH4 * 128;
    xr += l32 * 4;
    f32x4 v[NCH4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH4; ++c) {
        v[c] = *reinterpret_cast<const f32x4*>(xr + c * 128);
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float mean = half_wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = v[c][j] - mean;
            q += a * a;
        }
    const float rstd = 1.0f / sqrtf(half_wave_sum(q) * (1.0f / D) + eps);
    sh += l32 * 4;
    sc += l32 * 4;
    orow += l32 * 4;
#pragma unroll
    for (int c = 0; c < NCH4; ++c) {
        const V4 s4 = *reinterpret_cast<const V4*>(sc + c * 128);
        const V4 h4 = *reinterpret_cast<const V4*>(sh + c * 128);
        V4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m1 = rnd16<DT>(1.0f + (float)s4[j]);  // (1 + scale) is formed in the 16-bit type
            o[j] = (S)((v[c][j] - mean) * rstd * m1 + (float)h4[j]);
        }
        *reinterpret_cast<V4*>(orow + c * 128) = o;
    }
}
