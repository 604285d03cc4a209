// ConvTranspose3d(256 -> 256, kernel 2, stride 2) from the 4^3 to the 8^3 grid - the decoder's only upsample
// (models/vae3d_dib.py:250-261 UpBlock.upsample inside Decoder.forward) - with partial
// GroupNorm statistics of the NEXT layer (norm1 of up_blocks[1].nets[0]: 32 groups of 8 channels over the 512 output
// voxels) taken from the accumulators on the way out.
//
// A stride-2 kernel-2 transposed convolution has no overlap: output voxel (2z+dz, 2y+dy, 2x+dx) = W[:, :, dz,dy,dx]^T x
// in[z,y,x] + b, i.e. eight independent [64 voxels] x [256 -> 256] products per primitive.  As one GEMM (gemm.hip,
// EPI_CONVT: M = P * 64, N = 8 * 256, K = 256) it has FOUR k-tiles - all prologue and epilogue: 16,384 workgroups each
// loading 128 KB of operands for 32 KB of output (2.1 GB through L2 -> LDS, 344 us).
//
// WEIGHT-STATIONARY: a workgroup owns ONE tap and walks primitives; wave = (z-half of the primitive, 64 output
// channels) keeps ITS slice of the tap's matrix - 64 cout x 256 cin = 32 fragments - in 128 VGPRs for the whole kernel.
// The activations of a primitive (64 voxels x 256 channels, 32 KB) come through a 4-deep LDS ring filled by LDS-DMA:
// 64 MFMAs 16x16x32 per wave per primitive against 16 ds_read_b128, bias + rounding, four 16-byte stores.
// How it got here (2048 primitives; GEMM form 345-375 us):
//  * activation-stationary, weights streaming through an LDS-DMA ring: 286 us.  gfx9's vmcnt is ONE in-order counter for
//    loads, LDS-DMA and stores: every counted wait for a weight tile issued after a tap's output stores also waited for
//    those stores, two tile periods later;
//  * weight in LDS (128 KB), activations straight from global memory into registers in MFMA fragment layout with a ping-pong
//    prefetch: 300-320 us.  Probes: without the output stores 261, without the activation loads 155, with an L2-resident
//    1 MB activation set 326 - not HBM, not L2 misses, but the REQUEST rate: a `global_load_dwordx4` in fragment layout
//    has consecutive lanes on consecutive ROWS (512 B apart) = 64 separate 16-byte requests per instruction (128 of them
//    per primitive-tap per CU: ~10k cycles against 2k of MFMA).  (Also: a prefetch under `if` made hipcc's waitcnt pass
//    merge the two paths and wait as if the newer loads did not exist - the prefetches must be unconditional.)
//  * this version: the DMA's per-lane source addresses put 8 lanes on each 128-byte line (8 requests per instruction), the
//    ring is deep enough that an output store has three steps before a counted wait reaches it: **182 us** (754 TFLOP/s;
//    the 537 MB of output alone are 78 us at the 6.9 TB/s a pure fill reaches).
// Statistics: per lane shifted sums (x - shift, (x - shift)^2 of the ROUNDED 16-bit outputs, shift = the group's first bias)
// over the wave's 32 voxels, a 16-lane reduction, one (s1, s2) pair per (primitive, tap, z-half, group) written to
// `part[P][16][32][2]`; the consumer adds the 16 pairs in fixed order (deterministic; no atomics).
#include <stdlib.h>

#include "common.h"

namespace {

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// cout (0..255) held by LDS row rho = 64 g + 16 ni + i16: a lane (lg) then owns channels 64 g + 16 lg + 4 ni + r
__host__ __device__ __forceinline__ int cout_of_row(int rho) {
    const int i16 = rho & 15, ni = (rho >> 4) & 3, g = rho >> 6;
    return g * 64 + (i16 >> 2) * 16 + ni * 4 + (i16 & 3);
}

template <int DT>
__global__ __launch_bounds__(512) void convt_s4c256_kernel(const typename T16<DT>::S* __restrict__ in,
                                                          const typename T16<DT>::S* __restrict__ Wp,
                                                          const typename T16<DT>::S* __restrict__ bias,
                                                          typename T16<DT>::S* __restrict__ out, float* __restrict__ part, int P,
                                                          int ngroups) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    typedef __attribute__((address_space(3))) void LV;
    constexpr int CIN = 256, COUT = 256, KT = 256 * 64;               // halves per weight k-tile image [256 cout][64 k]
    constexpr int NST = 4, AST = 64 * 256;                            // activation ring: 4 primitives x [4 k-tiles][64 voxels][64 ch] (32 KB each)
    __shared__ __attribute__((aligned(16))) S ring[NST * AST];        // 128 KB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wz = wave >> 2, wn = wave & 3;                          // z-half of the primitive, group of 64 output channels
    const int lr = lane & 15, lg = lane >> 4;
    // workgroups are dealt to the 8 XCDs round-robin: the eight taps of a primitive group share blockIdx & 7, i.e. one XCD's
    // L2, and walk the same primitives at the same time (needs a multiple of 8 groups; otherwise the plain order)
    const bool xcd = (ngroups & 7) == 0;
    const int tap = xcd ? (blockIdx.x >> 3) & 7 : blockIdx.x & 7;
    const int grp = xcd ? (blockIdx.x & 7) | ((blockIdx.x >> 6) << 3) : blockIdx.x >> 3;
    const int dz = tap >> 2, dy = (tap >> 1) & 1, dx = tap & 1;
    const int nstep = grp < P ? (P - grp + ngroups - 1) / ngroups : 0;   // primitives grp, grp + ngroups, ...

    // ---- activation DMA: primitive n -> stage n & 3.  Instruction i of this wave copies rows 8 wave .. + 8 of k-tile i: the
    // lane's 16 bytes are chunk (lane & 7) ^ swizzle of row 8 wave + (lane >> 3) - 8 lanes per 128-byte line, where a load
    // in MFMA fragment layout would be one 16-byte request per lane (this kernel's first version: 10k cycles per step)
    const int drow = 8 * wave + (lane >> 3);
    const unsigned voff = (unsigned)((drow * CIN + (((lane & 7) ^ ((drow >> 1) & 7)) << 3)) * 2);      // bytes inside the primitive
    const unsigned lds0 = (unsigned)(uintptr_t)(LV*)ring + (unsigned)wave * 1024u;
    auto issue = [&](int n) {
        const int p = min(grp + n * ngroups, P - 1);                  // (past the end: a harmless reload of the last primitive)
        const char* sb = reinterpret_cast<const char*>(in) + (int64_t)__builtin_amdgcn_readfirstlane(p) * (64 * CIN * 2);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(n & 3) * (AST * 2));
        // (the instruction offset of an LDS-DMA load is added to the LDS address as well as to the global one: k-tile i is 128 i
        // bytes further in memory and 8 KB i further in LDS, so M0 advances by 0x2000 - 0x80 per instruction)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_add_u32 m0, m0, 0x1f80\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:128\n\t"
                     "s_add_u32 m0, m0, 0x1f80\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:256\n\t"
                     "s_add_u32 m0, m0, 0x1f80\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:384"
                     ::"s"(m0v), "v"(voff), "s"(sb) : "memory");
    };
    issue(0);
    issue(1);
    issue(2);

    // ---- this wave's weights: 64 output channels x 256 k of the tap = 32 fragments = 128 registers, read once from the packed
    // image (same offsets as an LDS read of it would use)
    V8 wreg[8][4];
    {
        const S* wsrc = Wp + (int64_t)tap * 4 * KT;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                wreg[ks][ni] = *reinterpret_cast<const V8*>(wsrc + (ks >> 1) * KT + lds_off(wn * 64 + ni * 16 + lr, (ks & 1) * 4 + lg));
    }
    const int c0 = wn * 64 + lg * 16;                                 // this lane's 16 output channels = groups c0 / 8, c0 / 8 + 1
    V8 b0 = V8{}, b1 = V8{};
    if (bias) { b0 = *reinterpret_cast<const V8*>(bias + c0); b1 = *reinterpret_cast<const V8*>(bias + c0 + 8); }
    const float sh0 = (float)b0[0], sh1 = (float)b1[0];
    const int vy = lr >> 2, vx = lr & 3;                              // in-plane position of this lane's input voxel
    // all of the above "used" here, so that hipcc waits for these loads once, before the loop (conv3s8.hip)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) asm volatile("" ::"v"(wreg[ks][ni]));
    asm volatile("" ::"v"(b0), "v"(b1));

#pragma clang loop unroll(disable)
    for (int n = 0; n < nstep; ++n) {
        // Primitive n landed for this wave.  vmcnt is ONE in-order counter: behind D(n) [issued at step n - 3] sit S(n-3), D(n+1),
        // S(n-2), D(n+2), S(n-1) - two groups of 4 DMAs and three groups of 4 (+ 1 with statistics) stores may stay in flight,
        // so an output store has three steps to complete.
        if (part) asm volatile("s_waitcnt vmcnt(23)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(20)\n\ts_barrier" ::: "memory");
        issue(n + 3);                                                 // into the stage of primitive n - 1, which every wave has left
        const int p = grp + n * ngroups;
        const S* As = ring + (n & 3) * AST;
        f32x4 acc[2][4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            V8 af[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                af[mi] = *reinterpret_cast<const V8*>(As + (ks >> 1) * (64 * 64) + lds_off(wz * 32 + mi * 16 + lr, (ks & 1) * 4 + lg));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = T16<DT>::mfma16(wreg[ks][ni], af[mi], acc[mi][ni]);
        }
        // ---- 32 voxels x 64 channels: bias, rounding, store at (2z+dz, 2y+dy, 2x+dx), statistics
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int v8 = ((2 * (2 * wz + mi) + dz) * 8 + 2 * vy + dy) * 8 + 2 * vx + dx;
            V8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o0[e] = (S)(acc[mi][e >> 2][e & 3] + (float)b0[e]);
                o1[e] = (S)(acc[mi][2 + (e >> 2)][e & 3] + (float)b1[e]);
                const float d0 = (float)o0[e] - sh0, d1 = (float)o1[e] - sh1;
                s1[0] += d0; s2[0] += d0 * d0;
                s1[1] += d1; s2[1] += d1 * d1;
            }
            S* dst = out + ((int64_t)p * 512 + v8) * COUT + c0;
            *reinterpret_cast<V8*>(dst) = o0;
            *reinterpret_cast<V8*>(dst + 8) = o1;
        }
        if (part) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    s1[k] += __shfl_xor(s1[k], off);
                    s2[k] += __shfl_xor(s2[k], off);
                }
            }
            if (lr == 0) {                                            // part[p][tap * 2 + wz][group][2], group = wn * 8 + 2 lg + k
                float* dst = part + (((int64_t)p * 16 + tap * 2 + wz) * 32 + wn * 8 + 2 * lg) * 2;
                *reinterpret_cast<f32x4*>(dst) = f32x4{s1[0], s2[0], s1[1], s2[1]};
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the tail DMAs must not outlive the workgroup's LDS
}

// Wt [8 * 256][256] (row = tap * 256 + co, K = ci) -> Wp[tap][kc][rho][slot][8]; one 16-byte chunk per thread
__global__ __launch_bounds__(256) void convt_s4_pack_kernel(const unsigned short* __restrict__ Wt, unsigned short* __restrict__ Wp) {
    const int cid = blockIdx.x * 256 + threadIdx.x;          // chunk index in Wp: ((tile * 256 + rho) * 8 + slot)
    if (cid >= 8 * 4 * 256 * 8) return;
    const int slot = cid & 7, rho = (cid >> 3) & 255, tile = cid >> 11;
    const int kc = tile & 3, tap = tile >> 2;
    const int c = slot ^ ((rho >> 1) & 7);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<u4*>(Wp + (int64_t)cid * 8) =
        *reinterpret_cast<const u4*>(Wt + ((int64_t)tap * 256 + cout_of_row(rho)) * 256 + kc * 64 + c * 8);
}

}  // namespace

extern "C" int primx_convtranspose_s4_pack(const void* Wt, void* Wp, int dtype, void* stream) {
    PRIMX_REQUIRE(Wt && Wp && Wt != Wp, "primx_convtranspose_s4_pack: null or aliased pointer");
    PRIMX_REQUIRE(dtype == PRIMX_F16 || dtype == PRIMX_BF16, "primx_convtranspose_s4_pack: dtype must be PRIMX_F16 or PRIMX_BF16");
    hipLaunchKernelGGL(convt_s4_pack_kernel, dim3(8 * 4 * 256 * 8 / 256), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)Wt, (unsigned short*)Wp);
    PRIMX_CHECK_LAUNCH("primx_convtranspose_s4_pack");
    return PRIMX_OK;
}

extern "C" int primx_convtranspose_s4_packed(const void* in, const void* Wp, const void* bias, void* out, float* part, int P,
                                             int dtype, void* stream) {
    PRIMX_REQUIRE(in && Wp && out, "primx_convtranspose_s4_packed: null pointer");
    PRIMX_REQUIRE(P > 0, "primx_convtranspose_s4_packed: need P > 0 (P=%d)", P);
    static const int n_cu = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    int ngroups = n_cu / 8 > 0 ? n_cu / 8 : 1;                         // one workgroup per CU: 8 taps x n_cu / 8 primitive groups
    if (ngroups > P) ngroups = P;
    PRIMX_DISPATCH_16(dtype, "primx_convtranspose_s4_packed", {
        using Sx = typename T16<DT>::S;
        hipLaunchKernelGGL((convt_s4c256_kernel<DT>), dim3(8 * ngroups), dim3(512), 0, (hipStream_t)stream, (const Sx*)in,
                           (const Sx*)Wp, (const Sx*)bias, (Sx*)out, part, P, ngroups);
    });
    PRIMX_CHECK_LAUNCH("primx_convtranspose_s4_packed");
    return PRIMX_OK;
}
