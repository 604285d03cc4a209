// Shared device/host helpers for libprimx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/primx_hip.h"

// ------------------------------------------------------------------ error plumbing
void primx_set_error(const char* fmt, ...);

#define PRIMX_REQUIRE(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            primx_set_error(__VA_ARGS__); \
            return PRIMX_EINVAL;          \
        }                                 \
    } while (0)

#define PRIMX_CHECK_LAUNCH(name)                                                   \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            primx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return PRIMX_ELAUNCH;                                                  \
        }                                                                          \
    } while (0)

// ------------------------------------------------------------------ vector types
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// 16-bit storage type traits.  DT = PRIMX_F16 or PRIMX_BF16.
template <int DT>
struct T16;

template <>
struct T16<PRIMX_F16> {
    using S = _Float16;
    using V8 = f16x8;
    using V4 = f16x4;
    using V2 = f16x2;
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

template <>
struct T16<PRIMX_BF16> {
    using S = __bf16;
    using V8 = bf16x8;
    using V4 = bf16x4;
    using V2 = bf16x2;
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

// round-trip through the 16-bit type (the rounding autocast applies after every op)
template <int DT>
__device__ __forceinline__ float rnd16(float v) {
    return (float)(typename T16<DT>::S)v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

// tanh-approximated GELU as torch evaluates it in fp32: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
__device__ __forceinline__ float gelu_tanh_f(float x) {
    // 0.5 x (1 + tanh(u)) = x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3): one v_exp_f32 + one v_rcp_f32 + 5 VALU
    // instead of libm's tanhf (~40 instructions with range branches; 144 of them per thread made up a third of the
    // fc1 kernel, PRIMX_GEMM_PROF).  exp2 overflow -> inf -> result 0 (x << 0), underflow -> result x: both limits
    // are exact; elsewhere the error is a few fp32 ulp, far below the 16-bit rounding that follows.
    const float kC1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
    const float kC2 = kC1 * 0.044715f;
    const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(kC2, x * x, kC1));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// Four values at once in PACKED fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per issue slot): the
// same operations in the same order as gelu_tanh_f, so the results are the same to the last bit; 2 + 4 x 0.5 issue slots per value
// instead of 2 + 5.  The fc1 epilogue is bound by exactly this arithmetic (144 values per lane, two waves per SIMD: round 6,
// profiles/r6_epilogue_valu.txt).
__device__ __forceinline__ f32x4 gelu_tanh4(const f32x4 x) {
    const float kC1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float kC2 = kC1 * 0.044715f;
    const f32x4 a = x * __builtin_elementwise_fma(f32x4{kC2, kC2, kC2, kC2}, x * x, f32x4{kC1, kC1, kC1, kC1});
    f32x4 e;
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(a[j]);
    e = e + 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_rcpf(e[j]);
    return x * e;
}

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }

// Dispatch a 16-bit dtype code to a template instantiation.
#define PRIMX_DISPATCH_16(dtype, NAME, ...)                                  \
    do {                                                                     \
        if ((dtype) == PRIMX_F16) {                                          \
            constexpr int DT = PRIMX_F16;                                    \
            __VA_ARGS__;                                                     \
        } else if ((dtype) == PRIMX_BF16) {                                  \
            constexpr int DT = PRIMX_BF16;                                   \
            __VA_ARGS__;                                                     \
        } else {                                                             \
            primx_set_error("%s: dtype must be PRIMX_F16 or PRIMX_BF16", NAME); \
            return PRIMX_EINVAL;                                             \
        }                                                                    \
    } while (0)

// Row stride (elements) of the token-major head layouts: PRIMX_HEADS_ROWS = DP, PRIMX_HEADS_KROWS = DP + 8.
__host__ __device__ __forceinline__ int heads_row_stride(int kind, int DP) { return kind == PRIMX_HEADS_KROWS ? DP + 8 : DP; }

// Position of key k inside its group of 16 in the PRIMX_HEADS_VT layout: the 4-key quads are
// stored in the order {0,2,1,3} so that the lane half (hi) of a 32x32x16 MFMA reads the 8 keys
// its own accumulator registers hold as one contiguous 16-byte vector (see attention.hip).
__host__ __device__ __forceinline__ int vt_key_pos(int k) {
    int quad = (k >> 2) & 3;
    int perm = ((quad & 1) << 1) | (quad >> 1);
    return (k & ~15) | (perm << 2) | (k & 3);
}
