// 3x3x3 / stride 1 / pad 1 convolution of the VAE decoder's 4^3 stage (256 channels in, a multiple of 256 out) with the
// ACTIVATIONS HELD IN REGISTERS - the eight ResNet convolutions that are 56 % of the decode (models/vae3d_dib.py:62-75
// inside Decoder.forward :251-277).
//
// Why not the implicit GEMM of gemm.hip (GATHER = 1): there the 27 taps re-read every activation 27 times through
// global -> registers -> LDS -> registers, and a 128 x 128 tile with 64 x 64 wave tiles needs as many LDS-read cycles as
// MFMA cycles.  Here a quarter of a primitive's 4^3 x 256 input (64 channels) is 8 KB = 32 VGPRs per lane of ONE wave, laid
// out as the MFMA's voxel-side operand: fragment [ks][z] holds z-plane z (16 voxels = the 16 operand rows, lane & 15 = 4 y + x)
// and channels 64 q + 32 ks + 8 (lane >> 4) .. + 8.  A tap (dz, dy, dx) is then
//   dz: which fragment is multiplied (plane z + dz; a plane outside the volume is SKIPPED - 17 % fewer MFMAs than the
//       zero-padded GEMM),
//   dy, dx: a DPP row shift by 4 dy + dx lanes inside each 16-lane row (row_shl / row_shr with bound_ctrl: lanes shifted in
//       from outside the plane read 0) fused with the AND that masks the x wrap - one v_and_b32_dpp per register, 4 per
//       8 MFMAs.
// LDS carries only the weights: a 4-stage ring of [256 cout][64 k] tiles filled by LDS-DMA (global_load_lds).
//
// Workgroup = 8 waves = 4 primitives x 2 column groups of 128 cout; wave tile 64 voxels x 128 cout = 32 accumulators of
// v_mfma_f32_16x16x32 with the operands SWAPPED (A = weights, B = voxels), so a lane ends up with 32 CONSECUTIVE output
// channels of one voxel.  (First version: 2 primitives x 4 groups of 64 cout with all 256 channels resident.  PRIMX_CONV_PROF
// showed it bound by the weight stream - one 32 KB tile per 128 output rows, 19.6 B/clk/CU through the L2 -> LDS path, the
// MFMAs entirely hidden; four primitives per workgroup halve that stream.  Two waves per SIMD leave 256 registers per
// lane, 128 of them accumulators and 32 weight fragments; with 64 activation registers (channel halves) hipcc spilled 520
// VGPRs, hence channel QUARTERS: the loop runs all 27 taps for channels 0..63, reloads the 32 activation registers, runs
// them for 64..127, and so on - each reload is 8 loads per lane whose latency is exposed, three times per workgroup.)
//
// Weights are PRE-PACKED once (primx_conv3d_s4_pack) into the exact LDS image of every tile, so that a wave's DMA
// instruction copies 1 KB of consecutive global memory: Wp[cout block][tap][64-channel group kc][LDS row rho][slot][8], where
//   LDS row rho = 128 g + 16 ni + i16  holds cout  128 g + 32 (i16 >> 2) + 4 ni + (i16 & 3)   (operand-row permutation that
//       gives a lane consecutive channels), and
//   slot s of row rho holds k-chunk  s ^ ((rho >> 1) & 7)   (the bank swizzle of gemm.hip's lds_off).
// (The row-major [cout][6912] layout put a tile's 256 row segments 13,824 bytes apart: 54 x 256 B, i.e. on 8 of the 16 L2
// channels.)  Tile order: for channel quarter q [dynamic, 4] for (dy,dx) [static, 9] for dz [static, 3]: tile index
// 27 q + 3 dydx + dz, ring stage = index & 3 = (static part + 3 q) & 3.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// cout (0..255 inside the block) held by LDS row rho
__host__ __device__ __forceinline__ int cout_of_row(int rho) {
    const int i16 = rho & 15, ni = (rho >> 4) & 7, g = rho >> 7;
    return g * 128 + (i16 >> 2) * 32 + ni * 4 + (i16 & 3);
}

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

// One 16x16x32 MFMA whose accumulator lives in the ACCUMULATOR file ("+a").  With the 128 accumulators in VGPRs the
// kernel sat at exactly 256 registers and hipcc spilled: three accumulator quads across every channel-quarter boundary
// (stores + reloads with s_waitcnt vmcnt(0) INSIDE the main loop, i.e. a drain of the weight DMA queue) and five address
// registers - 72 bytes of scratch per lane, 40 MB of extra writes per launch (round-2 PMC: WRITE_SIZE 108 MB for a 67 MB
// output).  The unified file is split per kernel: 128 AGPRs for the accumulators leave the 128 VGPRs to the activation
// quarter (32), the weight fragments (32), the shifted plane and addresses - no spill.  FIRST = the operand `b` was just
// written by the DPP shifts (VALU write -> MFMA read needs wait states that the compiler cannot see inside asm).
template <int DT, bool FIRST>
__device__ __forceinline__ void mfma16_acc(f32x4& acc, const typename T16<DT>::V8 a, const typename T16<DT>::V8 b) {
    if constexpr (DT == PRIMX_F16) {
        if constexpr (FIRST) asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    } else {
        if constexpr (FIRST) asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
}

// PRIMX_CONV_PROF=1 timeline (sums over all waves, core cycles): [0] waves, [1] entry -> tile 0 landed, [2] main loop,
// [3] epilogue, [4] of the main loop: parked at the per-tile wait + barrier, [5] weight-DMA issue
__device__ unsigned long long g_conv_prof[8];

template <int DT, int PROF>
__global__ __launch_bounds__(512) void conv3_s4c256_kernel(const typename T16<DT>::S* __restrict__ in,
                                                          const typename T16<DT>::S* __restrict__ Wp,
                                                          const typename T16<DT>::S* __restrict__ bias,
                                                          const typename T16<DT>::S* __restrict__ res, float res_scale,
                                                          typename T16<DT>::S* __restrict__ out, int P, int Cout) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    typedef __attribute__((address_space(3))) void LV;
    unsigned long long pc0 = 0, pc1 = 0, pc2 = 0, p_wait = 0, p_iss = 0;
    if (PROF) pc0 = __builtin_readcyclecounter();
    constexpr int CIN = 256, VOX = 64, NST = 4, STAGE = 256 * 64;   // halves per ring stage (32 KB)
    constexpr int NI = 8;
    __shared__ __attribute__((aligned(16))) S smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int ncb = Cout >> 8;
    const int quad = blockIdx.x / ncb, cb = blockIdx.x - quad * ncb;
    const int prim = quad * 4 + wp;
    const int prim_ld = min(prim, P - 1);           // P % 4 != 0: the surplus waves recompute primitive P-1 and store nothing

    // ---- one channel quarter of the primitive's activations: 8 fragments of 16 voxels x 32 channels
    V8 a[2][4];
    const S* a_src = in + ((int64_t)prim_ld * VOX + lr) * CIN + lg * 8;
    auto load_a = [&](int q) {
#pragma unroll
        for (int z = 0; z < 4; ++z)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a[ks][z] = *reinterpret_cast<const V8*>(a_src + z * 16 * CIN + q * 64 + ks * 32);
    };
    load_a(0);
    const int mask_xp = (lr & 3) != 3 ? -1 : 0;     // dx = +1: x = 3 has no right neighbour
    const int mask_xm = (lr & 3) != 0 ? -1 : 0;     // dx = -1: x = 0 has no left neighbour

    // ---- weight DMA: instruction I = wave + 8 i copies bytes [1024 I, 1024 I + 1024) of the packed tile to the same offset of
    // the stage (lane-linear on both sides)
    const char* wtiles = reinterpret_cast<const char*>(Wp) + (int64_t)cb * 27 * 4 * STAGE * 2;
    const unsigned voff = (unsigned)(wave * 1024 + lane * 16);   // byte offset of this lane's 16 bytes inside an 8 KB group of 8 pieces
    // (inline asm: for this address hipcc's global_load_lds builtin selects the 64-bit VGPR-address form - two more VGPRs per
    // instruction and a v_lshl_add_u64 each - where SGPR base + 32-bit lane offset does; M0 = LDS byte address of the piece)
    const unsigned lds0 = (unsigned)(uintptr_t)(LV*)smem + (unsigned)wave * 1024u;
    auto issue_one = [&](int q, int dydx, int dzi, int stage, int i) {
        const int tile = __builtin_amdgcn_readfirstlane((dzi * 9 + dydx) * 4 + q);
        const char* sb = wtiles + (int64_t)tile * (STAGE * 2) + i * 8192;
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)stage * (STAGE * 2) + (unsigned)i * 8192u);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sb) : "memory");
    };
    auto issue = [&](int q, int dydx, int dzi, int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_one(q, dydx, dzi, stage, i);
    };

    f32x4 acc[4][NI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // tiles 0, 1, 2 = (q 0, dydx 0, dz 0..2)
    issue(0, 0, 0, 0);
    issue(0, 0, 1, 1);
    issue(0, 0, 2, 2);

    const int w_row = wn * 128 + lr;
    // Weight fragments are read ONE K-STEP AHEAD, across the tile boundary too: the barrier at the top of tile t orders
    // "tiles <= t + 1 landed for every wave", so the first fragments of tile t + 1 can be requested right after the last
    // MFMAs of tile t have issued and their LDS latency passes under those MFMAs and the barrier wait.  The DMA of tile
    // t + 3 is issued at tile t and must have landed by the barrier of tile t + 2: two tile periods for an L2-resident tile.
    V8 wf[NI];
    auto read_wf = [&](int stage, int ksl) {
        const S* Ws = smem + stage * STAGE;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const V8*>(Ws + lds_off(w_row + ni * 16, ksl * 4 + lg));
    };
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");   // tile 0 landed
    if (PROF) pc1 = __builtin_readcyclecounter();
    read_wf(0, 0);

    for (int q = 0; q < 4; ++q) {
        // (every MFMA that reads the previous quarter has issued; the loads' latency is exposed.  Staging the next quarter
        // through the spare 32 KB of LDS by LDS-DMA a few tiles ahead - coalesced, prefetched - measured 3 % SLOWER same-box:
        // with four more DMAs in the in-order vmcnt queue the next counted waits stall longer than the exposed loads cost.)
        if (q > 0) load_a(q);
        const int sq = (3 * q) & 3;
        // one tap column (dy, dx) = 3 tiles; a generic lambda over an integral constant, called nine times: the DPP controls
        // are immediates, so the column index must be a compile-time constant (a 9-way switch per fragment measured as
        // 2,500 scalar branches; #pragma unroll refuses a body this large)
        auto column = [&](auto dydx_c) {
            constexpr int dydx = decltype(dydx_c)::value;
            constexpr int DY = dydx / 3 - 1, DX = dydx % 3 - 1;
#pragma unroll
            for (int dzi = 0; dzi < 3; ++dzi) {
                const int stage = (((3 * dydx + dzi) & 3) + sq) & 3;       // tile index 27 q + 3 dydx + dzi, mod 4
                // tiles <= t + 1 landed for this wave (tile t + 2 may stay in flight); every wave past this barrier has
                // consumed its fragments of tile t - 1, so the DMA below may overwrite that stage
                unsigned long long pa = 0, pb = 0;
                if (PROF) pa = __builtin_readcyclecounter();
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
                if (PROF) pb = __builtin_readcyclecounter();
                // tile + 3 = the same dz in the next tap column (of the next quarter after the last column); past the end: a
                // harmless reload of the last column.  Its four DMA instructions are issued one after each of the first four
                // plane groups below: issued together behind the barrier they stalled BOTH waves of a SIMD for ~500 cycles
                // per tile (PRIMX_CONV_PROF=1: 18 % of the main loop); spread out, the partner wave's MFMAs fill the stall.
                int qn = q, dn = dydx + 1;
                if (dn == 9) { dn = 0; qn = q + 1; }
                if (qn == 4) { qn = 3; dn = 8; }
                int n_dma = 0;
                if (PROF) p_wait += pb - pa;
#pragma unroll
                for (int ksl = 0; ksl < 2; ++ksl) {
                    // plane by plane: shift (4 VALU), then the 8 MFMAs of the output plane it feeds - one shifted fragment live
#pragma unroll
                    for (int z = 0; z < 4; ++z) {
                        const int mi = z + 1 - dzi;                    // output plane fed by source plane z under this dz
                        if (mi < 0 || mi > 3) continue;
                        // (opaque to CSE: the three dz tiles of a column shift the same fragments, and keeping those
                        // registers alive across tiles is what this kernel has no room for)
                        asm volatile("" : "+v"(a[ksl][z]));
                        const V8 sh = shift_plane<DY, DX>(a[ksl][z], mask_xp, mask_xm);
                        mfma16_acc<DT, true>(acc[mi][0], wf[0], sh);
#pragma unroll
                        for (int ni = 1; ni < NI; ++ni) mfma16_acc<DT, false>(acc[mi][ni], wf[ni], sh);
                        if (n_dma < 4) {
                            __builtin_amdgcn_sched_barrier(0);     // (keeps the DMA behind THIS plane group's MFMAs)
                            unsigned long long pi = 0;
                            if (PROF) pi = __builtin_readcyclecounter();
                            issue_one(qn, dn, dzi, (stage + 3) & 3, n_dma++);
                            if (PROF) p_iss += __builtin_readcyclecounter() - pi;
                        }
                    }
                    // (fenced: left free, hipcc hoists these reads into the MFMA groups above through a SECOND set of 32
                    // fragment registers, which this kernel does not have - it then spilled ~7 VGPR quads per tile)
                    __builtin_amdgcn_sched_barrier(0);
                    if (ksl == 0) read_wf(stage, 1);
                    else read_wf((stage + 1) & 3, 0);                 // (after the very last tile: a read nobody uses)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        column(std::integral_constant<int, 0>{}); column(std::integral_constant<int, 1>{}); column(std::integral_constant<int, 2>{});
        column(std::integral_constant<int, 3>{}); column(std::integral_constant<int, 4>{}); column(std::integral_constant<int, 5>{});
        column(std::integral_constant<int, 6>{}); column(std::integral_constant<int, 7>{}); column(std::integral_constant<int, 8>{});
    }
    // The MFMAs above are inline asm (mfma16_acc): the compiler's hazard recognizer does not see them, so the wait states between
    // the LAST matrix instruction's accumulator write and the first v_accvgpr_read of the epilogue are provided here explicitly
    // instead of by whatever happens to be scheduled in between (an 8-pass 16x16x32 XDL write needs 11 before a VALU / accvgpr read
    // of the same registers, MI300 ISA guide 4.5 "MFMA hazards": 24 are given), and nothing may move across the fence.
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the redundant tail DMAs must not outlive the workgroup's LDS
    if (PROF) pc2 = __builtin_readcyclecounter();
    auto prof_end = [&]() {
        if (PROF) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long pc3 = __builtin_readcyclecounter();
            if (lane == 0) {
                atomicAdd(&g_conv_prof[0], 1ull); atomicAdd(&g_conv_prof[1], pc1 - pc0); atomicAdd(&g_conv_prof[2], pc2 - pc1);
                atomicAdd(&g_conv_prof[3], pc3 - pc2); atomicAdd(&g_conv_prof[4], p_wait); atomicAdd(&g_conv_prof[5], p_iss);
            }
        }
    };

    // ---- epilogue.  A lane holds voxel 16 mi + lr and channels 32 lg + 4 ni + r of its wave's 64 x 128 tile: stored from
    // there, one instruction touches sixteen 512-byte rows with four scattered 16-byte pieces each, and the store tail
    // measured 76k cycles per workgroup (PRIMX_CONV_PROF=1; 19 % of the kernel).  Instead each wave transposes through its own
    // 16 KB of the (now idle) ring, two output planes at a time as fp32, and walks the rows with 16 lanes per row: 256
    // contiguous bytes per row for the residual load and for the store.  LDS row = 32 floats x 4 ... [row][128 floats] with
    // the 16-byte chunk index XOR (row & 7): conflict-free for the 8-lane ds_write_b128 groups and the ds_read_b128 groups.
    __syncthreads();                                   // every wave is done with the ring
    float* stg = reinterpret_cast<float*>(smem) + wave * 4096;
    const int rr = lane >> 4, cc = lane & 15;          // row-major walk: row 4 i + rr, channels 8 cc .. + 8
    const int nb = cb * 256 + wn * 128 + 8 * cc;
    // (two straight-line variants, all 16 residual loads first: with the loads inside the row loop hipcc waited vmcnt(0)
    // per row, which on gfx9 also waits for the previous row's STORE - sixteen serialized store round trips per wave.
    // No buffering of the converted rows: that version spilled 30 VGPRs per lane = 31 MB of scratch writes per launch,
    // visible as WRITE_SIZE 110 MB against 67 MB of output)
    auto epilogue = [&](auto has_res_c) {
        constexpr bool HAS_RES = decltype(has_res_c)::value;
        const int64_t off0 = ((int64_t)prim * VOX + rr) * Cout + nb;
        // all 16 residual rows first (64 registers, the accumulators' 128 are still live; fragments and weights are dead)
        V8 rq[2][8];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
                for (int i = 0; i < 8; ++i) rq[p2][i] = *reinterpret_cast<const V8*>(res + off0 + (int64_t)(p2 * 32 + 4 * i) * Cout);
        }
        V8 bv = V8{};
        if (bias) bv = *reinterpret_cast<const V8*>(bias + nb);
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = m2 * 16 + lr, chunk = (lg * 8 + ni) ^ (row & 7);
                    *reinterpret_cast<f32x4*>(stg + row * 128 + chunk * 4) = acc[2 * p2 + m2][ni];
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 4 * i + rr;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + row * 128 + ((2 * cc) ^ (row & 7)) * 4);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + row * 128 + ((2 * cc + 1) ^ (row & 7)) * 4);
                V8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = (e < 4 ? v0[e] : v1[e - 4]) + (float)bv[e];
                    if constexpr (HAS_RES) y += (float)rq[p2][i][e];
                    o[e] = (S)(y * res_scale);
                }
                *reinterpret_cast<V8*>(out + off0 + (int64_t)(p2 * 32 + 4 * i) * Cout) = o;
            }
        }
    };
    if (prim < P) {
        if (res) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
    prof_end();
}

// Wk [Cout][27 * 256] (k = tap * 256 + ci) -> the packed tile images described at the top; one 16-byte chunk per thread.
template <typename S>
__global__ __launch_bounds__(256) void conv3_s4_pack_kernel(const S* __restrict__ Wk, S* __restrict__ Wp, int Cout) {
    const int64_t cid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // chunk index in Wp
    const int64_t total = (int64_t)Cout * 27 * 32;
    if (cid >= total) return;
    const int slot = (int)(cid & 7), rho = (int)((cid >> 3) & 255);
    const int64_t tile = cid >> 11;                                        // (cb * 27 + tap) * 4 + kc
    const int kc = (int)(tile & 3), tap = (int)((tile >> 2) % 27), cb = (int)((tile >> 2) / 27);
    const int n = cb * 256 + cout_of_row(rho);
    const int c = slot ^ ((rho >> 1) & 7);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<u4*>(Wp + cid * 8) = *reinterpret_cast<const u4*>(Wk + (int64_t)n * (27 * 256) + tap * 256 + kc * 64 + c * 8);
}

const int g_prof = [] {   // PRIMX_CONV_PROF=1: synchronous launches + timeline print
    const char* e = getenv("PRIMX_CONV_PROF");
    return e ? atoi(e) : 0;
}();

}  // namespace

extern "C" int primx_conv3d_s4_pack(const void* Wk, void* Wp, int Cout, int dtype, void* stream) {
    PRIMX_REQUIRE(Wk && Wp && Wk != Wp, "primx_conv3d_s4_pack: null or aliased pointer");
    PRIMX_REQUIRE(Cout > 0 && Cout % 256 == 0, "primx_conv3d_s4_pack: Cout must be a multiple of 256 (Cout=%d)", Cout);
    PRIMX_REQUIRE(dtype == PRIMX_F16 || dtype == PRIMX_BF16, "primx_conv3d_s4_pack: dtype must be PRIMX_F16 or PRIMX_BF16");
    const int64_t total = (int64_t)Cout * 27 * 32;
    hipLaunchKernelGGL((conv3_s4_pack_kernel<unsigned short>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)Wk, (unsigned short*)Wp, Cout);
    PRIMX_CHECK_LAUNCH("primx_conv3d_s4_pack");
    return PRIMX_OK;
}

extern "C" int primx_conv3d_s4_packed(const void* in, const void* Wp, const void* bias, const void* res, float res_scale,
                                      void* out, int P, int Cout, int dtype, void* stream) {
    PRIMX_REQUIRE(in && Wp && out, "primx_conv3d_s4_packed: null pointer");
    PRIMX_REQUIRE(P > 0 && Cout > 0 && Cout % 256 == 0, "primx_conv3d_s4_packed: need P > 0 and Cout %% 256 == 0 (P=%d Cout=%d)", P, Cout);
    PRIMX_DISPATCH_16(dtype, "primx_conv3d_s4_packed", {
        using Sx = typename T16<DT>::S;
        const dim3 grid(((P + 3) / 4) * (Cout / 256));
        if (g_prof == 1) {   // synchronous launch + timeline print
            unsigned long long z[8] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_prof), z, sizeof(z));
            hipLaunchKernelGGL((conv3_s4c256_kernel<DT, 1>), grid, dim3(512), 0, (hipStream_t)stream, (const Sx*)in, (const Sx*)Wp,
                               (const Sx*)bias, (const Sx*)res, res_scale, (Sx*)out, P, Cout);
            (void)hipStreamSynchronize((hipStream_t)stream);
            (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_conv_prof), sizeof(z));
            const double n = z[0] ? (double)z[0] : 1.0;
            fprintf(stderr, "conv3_s4c256<%d> P=%d Cout=%d: %llu waves; per wave (core cycles): entry->tile0 %.0f | main loop %.0f (parked at "
                            "wait+barrier %.0f, DMA issue %.0f; 108 tiles) | epilogue %.0f\n",
                    DT, P, Cout, z[0], z[1] / n, z[2] / n, z[4] / n, z[5] / n, z[3] / n);
        } else {
            hipLaunchKernelGGL((conv3_s4c256_kernel<DT, 0>), grid, dim3(512), 0, (hipStream_t)stream, (const Sx*)in, (const Sx*)Wp,
                               (const Sx*)bias, (const Sx*)res, res_scale, (Sx*)out, P, Cout);
        }
    });
    PRIMX_CHECK_LAUNCH("primx_conv3d_s4_packed");
    return PRIMX_OK;
}
