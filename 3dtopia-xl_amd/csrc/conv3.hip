// 3x3x3 / stride 1 / pad 1 convolution of the VAE decoder's 4^3 stage (256 channels in, a multiple of 256 out) with the
// ACTIVATIONS HELD IN REGISTERS - the eight ResNet convolutions that are 56 % of the decode (models/vae3d_dib.py:62-75
// inside Decoder.forward :251-277).
//
// Why not the implicit GEMM of gemm.hip (GATHER = 1): there the 27 taps re-read every activation 27 times through
// global -> registers -> LDS -> registers, and a 128 x 128 tile with 64 x 64 wave tiles needs as many LDS-read cycles as
// MFMA cycles.  Here a primitive's whole 4^3 x 256 input is 32 KB = 128 VGPRs per lane of ONE wave, laid out as the MFMA's
// voxel-side operand: fragment [ks][z] holds z-plane z (16 voxels = the 16 operand rows, lane & 15 = 4 y + x) and channels
// 32 ks + 8 (lane >> 4) .. + 8.  A tap (dz, dy, dx) is then
//   dz: which fragment is multiplied (plane z + dz; a plane outside the volume is SKIPPED - 17 % fewer MFMAs than the
//       zero-padded GEMM),
//   dy, dx: a DPP row shift by 4 dy + dx lanes inside each 16-lane row (row_shl / row_shr with bound_ctrl: lanes shifted in
//       from outside the plane read 0) and an AND with a per-lane mask for the x wrap - one VALU instruction per register,
//       16 per (tap column, 32 channels), shared by the three dz taps' 40 MFMAs.
// LDS carries only the weights: a 4-stage ring of [256 cout][64 k] tiles filled by LDS-DMA (global_load_lds), 4 KB of
// ds_read_b128 per wave per 16 MFMAs instead of 8 KB.
//
// Workgroup = 8 waves = 2 primitives x 4 column groups of 64 cout; wave tile 64 voxels x 64 cout = 16 accumulators of
// v_mfma_f32_16x16x32 with the operands SWAPPED (A = weights, B = voxels), so a lane ends up with 16 CONSECUTIVE output
// channels of one voxel (two 16-byte stores; the cout <-> operand-row permutation lives in the DMA's source addresses).
// Tile order: for (dy,dx) [dynamic, 9] for 64-channel group [static, 4] for dz [static, 3]; 108 tiles, 12 = 0 mod 4 so the
// ring stage of a tile is a compile-time constant.  The weight layout is the one conv3d_k3 always took:
// Wk[cout][k], k = tap * 256 + ci, tap = (dz * 3 + dy) * 3 + dx.
#include <type_traits>

#include "common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// one operand fragment (8 halves = 4 registers) moved by 4 DY + DX lanes inside each 16-lane row
template <int DY, int DX, typename V8>
__device__ __forceinline__ V8 shift_plane(const V8 v, const int mask_xp, const int mask_xm) {
    constexpr int SH = 4 * DY + DX;
    if constexpr (SH == 0) {
        return v;
    } else {
        constexpr int CTRL = SH > 0 ? 0x100 + SH : 0x110 - SH;   // row_shl:SH (lane i <- lane i + SH) / row_shr:-SH
        const i32x4 s = __builtin_bit_cast(i32x4, v);
        i32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (DX == 0) {
                d[e] = __builtin_amdgcn_update_dpp(0, s[e], CTRL, 0xF, 0xF, true);
            } else {
                // shift and x-wrap mask in ONE instruction (hipcc selects v_mov_b32_dpp + v_cndmask for the intrinsic form)
                const int m = DX > 0 ? mask_xp : mask_xm;
                int t;
                if constexpr (SH > 0)
                    asm("v_and_b32_dpp %0, %1, %2 row_shl:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(s[e]), "v"(m), "n"(SH));
                else
                    asm("v_and_b32_dpp %0, %1, %2 row_shr:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(s[e]), "v"(m), "n"(-SH));
                d[e] = t;
            }
        }
        return __builtin_bit_cast(V8, d);
    }
}

template <int DT>
__global__ __launch_bounds__(512) void conv3_s4c256_kernel(const typename T16<DT>::S* __restrict__ in,
                                                          const typename T16<DT>::S* __restrict__ Wk,
                                                          const typename T16<DT>::S* __restrict__ bias,
                                                          const typename T16<DT>::S* __restrict__ res, float res_scale,
                                                          typename T16<DT>::S* __restrict__ out, int P, int Cout, int Kpad) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    typedef __attribute__((address_space(1))) const void GV;
    typedef __attribute__((address_space(3))) void LV;
    constexpr int CIN = 256, VOX = 64, NST = 4, STAGE = 256 * 64;   // halves per ring stage (32 KB)
    __shared__ __attribute__((aligned(16))) S smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2, wn = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;
    const int ncb = Cout >> 8;
    const int pair = blockIdx.x / ncb, cb = blockIdx.x - pair * ncb;
    const int prim = pair * 2 + wp;
    const int prim_ld = min(prim, P - 1);           // odd P: the second half of the last pair recomputes P-1 and stores nothing

    // ---- the primitive's activations: 32 fragments of 16 voxels x 32 channels
    V8 a[8][4];
    {
        const S* src = in + ((int64_t)prim_ld * VOX + lr) * CIN + lg * 8;
#pragma unroll
        for (int z = 0; z < 4; ++z)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) a[ks][z] = *reinterpret_cast<const V8*>(src + z * 16 * CIN + ks * 32);
    }
    const int mask_xp = (lr & 3) != 3 ? -1 : 0;     // dx = +1: x = 3 has no right neighbour
    const int mask_xm = (lr & 3) != 0 ? -1 : 0;     // dx = -1: x = 0 has no left neighbour

    // ---- weight DMA: instruction I = wave + 8 i writes LDS rows 8 I .. 8 I + 7 of the stage (1 KB, lane-linear);
    // LDS row rho = 64 g + 16 ni + i16 holds cout 64 g + 16 (i16 >> 2) + 4 ni + (i16 & 3)
    // (instruction i covers operand group g = i, so the four sources differ by a uniform 64 rows: one 32-bit lane offset
    // and a scalar base per instruction - four 64-bit lane pointers cost 8 VGPRs this kernel does not have)
    unsigned voff;
    {
        const int rho = 8 * wave + (lane >> 3);                        // < 64
        const int i16 = rho & 15, ni = rho >> 4;
        const int n = (i16 >> 2) * 16 + ni * 4 + (i16 & 3);
        const int c = (lane & 7) ^ ((rho >> 1) & 7);
        voff = (unsigned)(n * Kpad + c * 8) * 2u;
    }
    const char* wbase = reinterpret_cast<const char*>(Wk + (int64_t)cb * 256 * Kpad);
    auto issue = [&](int dydx, int ks2, int dzi, int stage) {
        const int col0 = __builtin_amdgcn_readfirstlane(((dzi * 9 + dydx) * CIN) + ks2 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* sb = wbase + ((int64_t)i * 64 * Kpad + col0) * 2;       // uniform
            __builtin_amdgcn_global_load_lds((GV*)(uintptr_t)(sb + voff),
                                             (LV*)(smem + stage * STAGE + (wave + 8 * i) * 512), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // tiles 0, 1, 2 = (dydx 0, ks2 0, dz 0..2)
    issue(0, 0, 0, 0);
    issue(0, 0, 1, 1);
    issue(0, 0, 2, 2);

    const int w_row = wn * 64 + lr;
    // one tap column (dy, dx) = 12 tiles; a generic lambda over an integral constant, called nine times: the DPP controls
    // are immediates, so the column index must be a compile-time constant (a 9-way switch per fragment measured as 2,500
    // scalar branches; #pragma unroll refuses a body this large)
    auto column = [&](auto dydx_c) {
        constexpr int dydx = decltype(dydx_c)::value;
        constexpr int DY = dydx / 3 - 1, DX = dydx % 3 - 1;
#pragma unroll
        for (int ks2 = 0; ks2 < 4; ++ks2) {
#pragma unroll
            for (int dzi = 0; dzi < 3; ++dzi) {
                const int stage = (ks2 * 3 + dzi) & 3;                 // tile index = 12 dydx + 3 ks2 + dzi
                // tile landed for this wave (two later tiles may stay in flight); this wave's reads of the previous tile's
                // stage are complete, so after the barrier the DMA below may overwrite it
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                {   // tile + 3 = next 64-channel group of the same dz (wrapping to the next tap column); past the end: harmless reload
                    const int k2n = (ks2 + 1) & 3;
                    const int dn = min(dydx + (ks2 == 3 ? 1 : 0), 8);
                    issue(dn, k2n, dzi, (stage + 3) & 3);
                }
                const S* Ws = smem + stage * STAGE;
#pragma unroll
                for (int ksl = 0; ksl < 2; ++ksl) {
                    const int ks = 2 * ks2 + ksl;
                    V8 wf[4];
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        wf[ni] = *reinterpret_cast<const V8*>(Ws + lds_off(w_row + ni * 16, ksl * 4 + lg));
                    // plane by plane: shift (4 VALU), then the 4 MFMAs of the output plane it feeds - one shifted fragment live
#pragma unroll
                    for (int z = 0; z < 4; ++z) {
                        const int mi = z + 1 - dzi;                    // output plane fed by source plane z under this dz
                        if (mi < 0 || mi > 3) continue;
                        // (opaque to CSE: the three dz tiles of a column shift the same fragments, and keeping those 16 registers
                        // alive across tiles is what this kernel has no room for - it spilled 39 VGPRs)
                        asm volatile("" : "+v"(a[ks][z]));
                        const V8 sh = shift_plane<DY, DX>(a[ks][z], mask_xp, mask_xm);
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = T16<DT>::mfma16(wf[ni], sh, acc[mi][ni]);
                    }
                }
            }
        }
    };
    column(std::integral_constant<int, 0>{}); column(std::integral_constant<int, 1>{}); column(std::integral_constant<int, 2>{});
    column(std::integral_constant<int, 3>{}); column(std::integral_constant<int, 4>{}); column(std::integral_constant<int, 5>{});
    column(std::integral_constant<int, 6>{}); column(std::integral_constant<int, 7>{}); column(std::integral_constant<int, 8>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant tail DMAs must not outlive the workgroup's LDS

    // ---- epilogue: lane (lr, lg) holds voxel 16 mi + lr, channels n0 + 4 ni + r
    if (prim >= P) return;
    const int n0 = cb * 256 + wn * 64 + lg * 16;
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = 0.f;
    if (bias) {
        const V8 b0 = *reinterpret_cast<const V8*>(bias + n0), b1 = *reinterpret_cast<const V8*>(bias + n0 + 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { bv[e] = (float)b0[e]; bv[8 + e] = (float)b1[e]; }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t off = ((int64_t)prim * VOX + mi * 16 + lr) * Cout + n0;
        V8 r0 = V8{}, r1 = V8{};
        if (res) { r0 = *reinterpret_cast<const V8*>(res + off); r1 = *reinterpret_cast<const V8*>(res + off + 8); }
        V8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y0 = acc[mi][e >> 2][e & 3] + bv[e], y1 = acc[mi][2 + (e >> 2)][e & 3] + bv[8 + e];
            if (res) { y0 += (float)r0[e]; y1 += (float)r1[e]; }
            o0[e] = (S)(y0 * res_scale);
            o1[e] = (S)(y1 * res_scale);
        }
        *reinterpret_cast<V8*>(out + off) = o0;
        *reinterpret_cast<V8*>(out + off + 8) = o1;
    }
}

}  // namespace

// Called by primx_conv3d_k3 (gemm.hip) for S = 4, Cin = 256, Cout % 256 == 0, Kpad == 27 * 256.
int primx_conv3_s4c256_launch(const void* in, const void* Wk, const void* bias, const void* res, float res_scale, void* out,
                              int P, int Cout, int Kpad, int dtype, hipStream_t stream) {
    PRIMX_DISPATCH_16(dtype, "primx_conv3d_k3", {
        using Sx = typename T16<DT>::S;
        hipLaunchKernelGGL((conv3_s4c256_kernel<DT>), dim3(((P + 1) / 2) * (Cout / 256)), dim3(512), 0, stream,
                           (const Sx*)in, (const Sx*)Wk, (const Sx*)bias, (const Sx*)res, res_scale, (Sx*)out, P, Cout, Kpad);
    });
    PRIMX_CHECK_LAUNCH("primx_conv3d_k3");
    return PRIMX_OK;
}
