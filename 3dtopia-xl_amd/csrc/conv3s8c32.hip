// [GroupNorm(32 groups of 1 channel) + SiLU +] 3x3x3 / stride 1 / pad 1 convolution of 32 input channels on the 8^3 grid,
// [+ residual, * scale]: the decoder's 32-channel stage - conv2 of up_blocks[1].nets[0], both convolutions of nets[1],
// and norm_out + conv_out (models/vae3d_dib.py:62-75, 262-270, 366-367, 383-385).
//
// One primitive's 8^3 x 32 activations are 32 KB: the workgroup (8 waves, persistent over primitives) keeps them in LDS
// as a ZERO-HALOED volume - 10 x 10 x 16 rows of 64 bytes (x padded to 16 so that, with the 16-byte chunk index XOR
// ((row >> 1) & 3), every ds_read_b128 lane group of an operand fragment is bank-conflict free: brute-forced over all
// alignments) - next to the complete weight (27 taps x 32 cout x 32 cin = 54 KB), loaded once per workgroup.  A tap is
// then an ADDRESS offset: wave w accumulates output plane w in registers (4 column groups of 16 voxels x 32 cout),
// 27 x 8 MFMAs 16x16x32 with fragments read straight from the haloed volume; no gather, no masks, no DPP.  The implicit
// GEMM of gemm.hip re-gathered every activation 27 times from L1/L2 for a 32-column tile (380 TFLOP/s, 143 us per
// launch); this kernel is bound by reading the input and writing the output once (67 + 67 MB per launch).
//
// The GroupNorm in front of these convolutions has one channel per group, i.e. per-(primitive, channel) statistics over
// the 512 voxels the workgroup already holds: two passes over LDS (mean, then sum of squared deviations, as
// groupnorm_silu_reg_kernel does), y = silu(x * g * rstd + (b - mean * g * rstd)) rounded to 16 bits in place.  This
// removes four groupnorm_silu launches per decode and the 134 MB round trip of each normalised tensor.
//
// Weight image (primx_conv3d_s8c32_pack): Wp[tap][row][slot][8], row = ni * 16 + i16 holding cout
// (i16 >> 2) * 8 + ni * 4 + (i16 & 3) for 32 output channels (a lane then owns 8 consecutive channels of a voxel: 16-byte
// stores, 1 KB contiguous per store instruction) or cout = i16 for <= 16 output channels (conv_out: 6), slot s = k-chunk
// s ^ ((row >> 1) & 3).
#include <stdlib.h>

#include "common.h"

namespace {

template <int DT, int NI>
__global__ __launch_bounds__(512) void conv3_s8c32_kernel(const typename T16<DT>::S* __restrict__ in,
                                                         const typename T16<DT>::S* __restrict__ Wp,
                                                         const typename T16<DT>::S* __restrict__ bias,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         const typename T16<DT>::S* __restrict__ res, float res_scale,
                                                         typename T16<DT>::S* __restrict__ out, int P, int Cout) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    constexpr int CIN = 32, VOX = 512, NROW = 10 * 10 * 16, WROWS = 27 * NI * 16;
    __shared__ __attribute__((aligned(16))) S vol[NROW * 32];        // 102,400 B
    __shared__ __attribute__((aligned(16))) S wl[WROWS * 32];        // 55,296 B (NI = 2)
    __shared__ float part[16][32];                                   // per-(voxel slice, channel) partial statistics
    __shared__ float fin[2][32];                                     // per-channel scale and shift of the normalisation

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave = output z-plane
    const int j = lane & 15, q = lane >> 4;
    const bool gn = gamma != nullptr;

    // ---- once per workgroup: zero the volume (the halo stays zero), copy the weight image
    for (int c = tid; c < NROW * 4; c += 512) reinterpret_cast<u32x4*>(vol)[c] = u32x4{0u, 0u, 0u, 0u};
    for (int c = tid; c < WROWS * 4; c += 512) reinterpret_cast<u32x4*>(wl)[c] = reinterpret_cast<const u32x4*>(Wp)[c];

    // this thread's voxel (z = w, y, x) and its row in the haloed volume
    const int vy = (tid >> 3) & 7, vx = tid & 7;
    const int myrow = ((w + 1) * 10 + vy + 1) * 16 + vx + 1;
    S* myp = vol + myrow * 32;
    const int mysw = (myrow >> 1) & 3;

    V8 raw[4];
    auto load_raw = [&](int p) {
        const S* src = in + ((int64_t)p * VOX + tid) * CIN;
#pragma unroll
        for (int c = 0; c < 4; ++c) raw[c] = *reinterpret_cast<const V8*>(src + 8 * c);
    };
    int p = blockIdx.x;
    if (p < P) load_raw(p);

    for (; p < P; p += gridDim.x) {
        __syncthreads();                                             // the previous primitive's fragment reads are done (and the init above)
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<V8*>(myp + ((c ^ mysw) << 3)) = raw[c];
        if (gn) {
            // statistics: thread = (channel c = tid & 31, slice s = tid >> 5 of 32 voxels)
            const int ch = tid & 31, sl = tid >> 5;
            auto at = [&](int v) -> float {                          // activation of voxel v, channel ch
                const int r = (((v >> 6) + 1) * 10 + ((v >> 3) & 7) + 1) * 16 + (v & 7) + 1;
                return (float)vol[r * 32 + ((((ch >> 3) ^ ((r >> 1) & 3)) << 3) | (ch & 7))];
            };
            __syncthreads();
            float s1 = 0.f;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) s1 += at(sl * 32 + i);
            part[sl][ch] = s1;
            __syncthreads();
            float mean = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) mean += part[s][ch];
            mean *= (1.0f / 512.0f);
            float s2 = 0.f;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) {
                const float d = at(sl * 32 + i) - mean;
                s2 += d * d;
            }
            __syncthreads();                                         // everyone has read part (means)
            part[sl][ch] = s2;
            __syncthreads();
            if (tid < 32) {
                float var = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) var += part[s][tid];
                const float r = 1.0f / sqrtf(var * (1.0f / 512.0f) + eps);
                const float g = gamma[tid] * r;
                fin[0][tid] = g;
                fin[1][tid] = beta[tid] - mean * g;
            }
            __syncthreads();
            // normalise + SiLU this thread's voxel from its registers, in place
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                V8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (S)silu_f((float)raw[c][e] * fin[0][8 * c + e] + fin[1][8 * c + e]);
                *reinterpret_cast<V8*>(myp + ((c ^ mysw) << 3)) = o;
            }
        }
        if (p + (int)gridDim.x < P) load_raw(p + gridDim.x);         // next primitive's rows arrive under the MFMA phase
        __syncthreads();

        // ---- 27 taps: fragment addresses are row offsets into the haloed volume
        f32x4 acc[4][NI];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[cg][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        // row of this lane's voxel of column group cg for the CENTRE tap: z = w, y = 2 cg + (j >> 3), x = j & 7
        const int row0 = ((w + 1) * 10 + (j >> 3) + 1) * 16 + (j & 7) + 1;
#pragma unroll 3
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
            const int toff = (dz * 10 + dy) * 16 + dx;
            V8 wf[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int r = (tap * NI + ni) * 16 + j;
                wf[ni] = *reinterpret_cast<const V8*>(wl + r * 32 + ((q ^ ((r >> 1) & 3)) << 3));
            }
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                const int r = row0 + cg * 32 + toff;
                const V8 xf = *reinterpret_cast<const V8*>(vol + r * 32 + ((q ^ ((r >> 1) & 3)) << 3));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[cg][ni] = T16<DT>::mfma16(wf[ni], xf, acc[cg][ni]);
            }
        }

        // ---- epilogue: lane (j, q) holds voxel w * 64 + cg * 16 + j and channels q * 8 + ni * 4 + r (NI = 2) / q * 4 + r (NI = 1)
        if constexpr (NI == 2) {
            V8 bv = V8{};
            if (bias) bv = *reinterpret_cast<const V8*>(bias + q * 8);
            V8 rq[4];
            const int64_t off0 = ((int64_t)p * VOX + w * 64 + j) * 32 + q * 8;
            if (res) {
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) rq[cg] = *reinterpret_cast<const V8*>(res + off0 + cg * 16 * 32);
            }
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                V8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = acc[cg][e >> 2][e & 3] + (float)bv[e];
                    if (res) y += (float)rq[cg][e];
                    o[e] = (S)(y * res_scale);
                }
                *reinterpret_cast<V8*>(out + off0 + cg * 16 * 32) = o;
            }
        } else {
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                const int64_t vrow = (int64_t)p * VOX + w * 64 + cg * 16 + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = q * 4 + r;
                    if (c < Cout) {
                        float y = acc[cg][0][r] + (bias ? (float)bias[c] : 0.f);
                        if (res) y += (float)res[vrow * Cout + c];
                        out[vrow * Cout + c] = (S)(y * res_scale);
                    }
                }
            }
        }
    }
}

// Wk [Cout][Kpad] (k = tap * 32 + ci, Kpad >= 864) -> Wp[tap][row][slot][8]; one 16-byte chunk per thread
__global__ __launch_bounds__(256) void conv3_s8c32_pack_kernel(const unsigned short* __restrict__ Wk, unsigned short* __restrict__ Wp,
                                                              int Cout, int Kpad, int NI) {
    const int cid = blockIdx.x * 256 + threadIdx.x;
    if (cid >= 27 * NI * 16 * 4) return;
    const int slot = cid & 3, row = (cid >> 2) % (NI * 16), tap = (cid >> 2) / (NI * 16);
    const int r = tap * NI * 16 + row;                                  // row index in the image (its swizzle)
    const int i16 = row & 15, ni = row >> 4;
    const int n = NI == 2 ? (i16 >> 2) * 8 + ni * 4 + (i16 & 3) : i16;
    const int c = slot ^ ((r >> 1) & 3);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 v = u4{0u, 0u, 0u, 0u};
    if (n < Cout) v = *reinterpret_cast<const u4*>(Wk + (int64_t)n * Kpad + tap * 32 + c * 8);
    *reinterpret_cast<u4*>(Wp + (int64_t)cid * 8) = v;
}

}  // namespace

extern "C" int primx_conv3d_s8c32_pack(const void* Wk, void* Wp, int Cout, int Kpad, int dtype, void* stream) {
    PRIMX_REQUIRE(Wk && Wp && Wk != Wp, "primx_conv3d_s8c32_pack: null or aliased pointer");
    PRIMX_REQUIRE((Cout == 32 || (Cout > 0 && Cout <= 16)) && Kpad >= 864 && Kpad % 8 == 0,
                  "primx_conv3d_s8c32_pack: Cout must be 32 or <= 16 and Kpad >= 864 (Cout=%d Kpad=%d)", Cout, Kpad);
    PRIMX_REQUIRE(dtype == PRIMX_F16 || dtype == PRIMX_BF16, "primx_conv3d_s8c32_pack: dtype must be PRIMX_F16 or PRIMX_BF16");
    const int NI = Cout == 32 ? 2 : 1;
    hipLaunchKernelGGL(conv3_s8c32_pack_kernel, dim3((27 * NI * 16 * 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)Wk, (unsigned short*)Wp, Cout, Kpad, NI);
    PRIMX_CHECK_LAUNCH("primx_conv3d_s8c32_pack");
    return PRIMX_OK;
}

extern "C" int primx_conv3d_s8c32_packed(const void* in, const void* Wp, const void* bias, const float* gamma, const float* beta,
                                         float eps, const void* res, float res_scale, void* out, int P, int Cout, int dtype,
                                         void* stream) {
    PRIMX_REQUIRE(in && Wp && out, "primx_conv3d_s8c32_packed: null pointer");
    PRIMX_REQUIRE(P > 0 && (Cout == 32 || (Cout > 0 && Cout <= 16)), "primx_conv3d_s8c32_packed: need P > 0 and Cout == 32 or <= 16 (P=%d Cout=%d)", P, Cout);
    PRIMX_REQUIRE((gamma == nullptr) == (beta == nullptr), "primx_conv3d_s8c32_packed: gamma and beta come together");
    static const int n_cu = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const dim3 grid(P < n_cu ? P : n_cu);
    PRIMX_DISPATCH_16(dtype, "primx_conv3d_s8c32_packed", {
        using Sx = typename T16<DT>::S;
        if (Cout == 32)
            hipLaunchKernelGGL((conv3_s8c32_kernel<DT, 2>), grid, dim3(512), 0, (hipStream_t)stream, (const Sx*)in, (const Sx*)Wp,
                               (const Sx*)bias, gamma, beta, eps, (const Sx*)res, res_scale, (Sx*)out, P, Cout);
        else
            hipLaunchKernelGGL((conv3_s8c32_kernel<DT, 1>), grid, dim3(512), 0, (hipStream_t)stream, (const Sx*)in, (const Sx*)Wp,
                               (const Sx*)bias, gamma, beta, eps, (const Sx*)res, res_scale, (Sx*)out, P, Cout);
    });
    PRIMX_CHECK_LAUNCH("primx_conv3d_s8c32_packed");
    return PRIMX_OK;
}
