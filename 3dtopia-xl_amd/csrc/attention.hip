// Flash-style attention forward for gfx950 on the pre-laid-out operands written by the projection
// GEMMs' epilogues (primx_linear_heads) - the replacement for xformers.ops.memory_efficient_attention.
//
// Geometry: workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows
// for the whole kernel and keeps them in the LANE dimension of both MFMAs:
//
//   S^T[key, q]  = mfma_32x32x16( A = K tile rows (key-major, from LDS),  B = Q^T (registers) )
//   O^T[d,   q] += mfma_32x32x16( A = V^T tile rows (d-major, from LDS),  B = P^T (registers) )
//
// so lane (q = lane & 31) holds, for ITS query, 16 scores per 32-key sub-tile and 16 output features
// per 32-feature tile: running max / sum / rescale are per-lane scalars and the only cross-lane
// traffic per KV tile is one exchange of the tile max between the two half-waves (lane ^ 32).
// The accumulator register r of S^T holds key (r&3) + 8*(r>>2) + 4*hi of its 32-key sub-tile; the
// PRIMX_HEADS_VT layout stores V^T with the 4-key quads of every 16 keys in the order {0,2,1,3}, which
// makes the 8 keys a lane owns in one 16-key MFMA step a contiguous 16-byte LDS read - P never
// leaves registers and needs no permute.
//
// KV tiles of 64 keys go global -> registers -> LDS (double-buffered, clamped branch-free loads, predicated
// stores, one barrier per tile).  K is staged ONE TILE AHEAD of V so that QK^T of tile j+1 (MFMA) is issued
// before the softmax of tile j (VALU) and the two pipes overlap inside every wave.  LDS rows are padded so
// that the 16-byte-slot stride is odd (K rows DP+8 halves, V^T rows 72 halves): ds_read_b128 conflict-free.
// Head dim 72 is zero-padded to DP = 80 for QK^T (5 k-steps) and to 96 output rows for PV (3 tiles):
// 22 MFMAs per 64 keys x 32 queries = 0.72 MFLOP issued for 0.59 MFLOP algorithmic (81.8 %).  The padding is put
// to work: column 72 of Q/K carries the key-padding mask and row 72 of V^T is all ones, so the MFMAs themselves
// deliver masked scores and the softmax denominator - the kernel is issue-bound on softmax VALU at this head dim
// (~230 non-MFMA instructions per 22 MFMAs per tile; profiles/r1_attn_pmc.txt), every removed instruction counts.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BQ = 128;   // query rows per workgroup
constexpr int BKV = 64;   // keys per tile
constexpr int VROW = BKV + 8;

template <typename V8>
__device__ __forceinline__ V8 ldg16(const void* ptr) {
    typedef __attribute__((address_space(1))) const V8 GV8;
    return *reinterpret_cast<GV8*>(reinterpret_cast<uintptr_t>(ptr));
}

// ABL != 0: measurement-only ablations (PRIMX_ATTN_ABL), results are wrong by design
template <int DT, int KSTEPS, int DTILES, int KMASK, int ABL = 0>
__global__ __launch_bounds__(256, 2) void attn_kernel(const typename T16<DT>::S* __restrict__ Qp,
                                                      const typename T16<DT>::S* __restrict__ Kp,
                                                      const typename T16<DT>::S* __restrict__ Vt,
                                                      typename T16<DT>::S* __restrict__ out, int H, int nq, int nq_pad,
                                                      int nkv, int nkv_pad, int dh, float c /* scale * log2(e) */, int stagger) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    using V4 = typename T16<DT>::V4;
    constexpr int DP = 16 * KSTEPS;
    constexpr int KROW = DP + 8;
    constexpr int VR = 32 * DTILES;            // V^T rows an MFMA A-operand can touch
    constexpr int KT = BKV * KROW;             // halves in one K tile
    constexpr int VT_ = VR * VROW;             // halves in one V^T tile
    constexpr int BUF = KT + VT_;
    constexpr int KCH = BKV * DP / 8, VCH = DP * 8;  // 16-byte chunks per tile
    constexpr int KIT = (KCH + 255) / 256, VIT = (VCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) S smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.x;  // (batch, head) fastest: with B*H % 8 == 0 all query tiles of a head share an XCD's L2
    const int q0 = blockIdx.y * BQ;
    const S* Kbase = Kp + (int64_t)bh * nkv_pad * DP;
    const S* Vbase = Vt + (int64_t)bh * DP * nkv_pad;

    // Phase stagger: the two workgroups that share a CU start together and would stay phase-locked (both in
    // their MFMA bursts, then both in their softmax VALU bursts - measured: per-tile time = MFMA + VALU of both
    // waves).  Delaying every second "generation" of workgroups by ~half a tile period lets one wave's MFMAs
    // overlap its SIMD partner's VALU work.
    if (stagger > 0 && (((blockIdx.y * gridDim.x + blockIdx.x) >> 8) & 1)) {
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(8);  // 8 * 64 cycles per iteration
    }

    // zero the V^T rows >= DP (they only feed discarded output rows, but keep them finite)
    if (VR > DP) {
        for (int i = tid; i < (VR - DP) * VROW; i += 256) {
            smem[KT + DP * VROW + i] = (S)0.f;
            smem[BUF + KT + DP * VROW + i] = (S)0.f;
        }
    }

    // Q^T B-operand fragments: lane (q = l31, hi) holds d = 16 s + 8 hi .. +7
    V8 qf[KSTEPS];
    {
        const S* qrow = Qp + ((int64_t)bh * nq_pad + q0 + wave * 32 + l31) * DP + hi * 8;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) qf[s] = ldg16<V8>(qrow + s * 16);
    }

    // ---- loader geometry (chunk ids past the tile are clamped for the load and skipped for the store)
    int k_goff[KIT], k_loff[KIT], v_loff[VIT];
    int64_t v_goff[VIT];
    bool k_on[KIT], v_on[VIT];
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
        const int ch = tid + 256 * i;
        k_on[i] = ch < KCH;
        const int cc = k_on[i] ? ch : KCH - 1;
        k_goff[i] = cc * 8;                                      // the K tile is one contiguous block
        k_loff[i] = (cc / (DP / 8)) * KROW + (cc % (DP / 8)) * 8;
    }
#pragma unroll
    for (int i = 0; i < VIT; ++i) {
        const int ch = tid + 256 * i;
        v_on[i] = ch < VCH;
        const int cc = v_on[i] ? ch : VCH - 1;
        v_goff[i] = (int64_t)(cc >> 3) * nkv_pad + (cc & 7) * 8;
        v_loff[i] = (cc >> 3) * VROW + (cc & 7) * 8;
    }
    const int ntiles = (nkv + BKV - 1) / BKV;
    auto load_k = [&](int j, V8 (&kr)[KIT]) {
        const S* kt = Kbase + (int64_t)min(j, ntiles - 1) * BKV * DP;
#pragma unroll
        for (int i = 0; i < KIT; ++i) kr[i] = ldg16<V8>(kt + k_goff[i]);
    };
    auto load_v = [&](int j, V8 (&vr)[VIT]) {
        const S* vt = Vbase + min(j, ntiles - 1) * BKV;
#pragma unroll
        for (int i = 0; i < VIT; ++i) vr[i] = ldg16<V8>(vt + v_goff[i]);
    };
    auto store_k = [&](int buf, V8 (&kr)[KIT]) {
        S* kb = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < KIT; ++i)
            if (k_on[i]) *reinterpret_cast<V8*>(kb + k_loff[i]) = kr[i];
    };
    auto store_v = [&](int buf, V8 (&vr)[VIT]) {
        S* vb = smem + buf * BUF + KT;
#pragma unroll
        for (int i = 0; i < VIT; ++i)
            if (v_on[i]) *reinterpret_cast<V8*>(vb + v_loff[i]) = vr[i];
    };

    f32x16 o[DTILES];
#pragma unroll
    for (int t = 0; t < DTILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    // ---- LDS fragment reads are BATCHED and issued early (ablation: with per-MFMA "ds_read, wait, mfma" the exposed
    // LDS latency was 46 % of the kernel, profiles/r1_attn_ablation.txt): all K fragments of a tile in one burst, all
    // V^T fragments in another, each consumed after ONE pinned wait while VALU work covers the latency.
    auto read_k = [&](int buf, V8 (&kf)[2][KSTEPS]) {
        const S* kb = smem + buf * BUF + l31 * KROW + hi * 8;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
                kf[kt][s] = (ABL == 5) ? qf[s] : *reinterpret_cast<const V8*>(kb + kt * 32 * KROW + s * 16);
    };
    auto read_v = [&](int buf, int half, V8 (&vf)[DTILES][2]) {   // key-steps 2*half, 2*half+1 of the tile
        const S* vb = smem + buf * BUF + KT + l31 * VROW + hi * 8 + half * 32;
#pragma unroll
        for (int t = 0; t < DTILES; ++t)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
                vf[t][k2] = (ABL == 5) ? qf[k2] : *reinterpret_cast<const V8*>(vb + t * 32 * VROW + k2 * 16);
    };
    auto fence_lds = [&]() {  // every ds_read issued so far has landed; keep the compiler from moving MFMAs above it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // S^T(tile t) = K(t) Q^T.  Keys >= nkv: when the head dim has a spare padded column (dh < DP, e.g. 72 -> 80) they
    // are masked BY THE OPERANDS (Q[:, dh] = 1, K[pad rows, dh] = -30000, see primx_hip.h) and no code is needed here;
    // otherwise (KMASK) the scores are overwritten.
    auto qk = [&](const V8 (&kf)[2][KSTEPS], int t, f32x16 (&sc)[2]) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                if (ABL == 4) { if (s == 0) sc[kt] = zero16; asm volatile("" :: "v"(kf[kt][s])); }
                else sc[kt] = T16<DT>::mfma32(kf[kt][s], qf[s], s == 0 ? zero16 : sc[kt]);  // shared zero C operand
            }
        if (KMASK && (t + 1) * BKV > nkv) {
            const int kbase = t * BKV + 4 * hi;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= nkv) sc[kt][r] = -1e30f;
                }
        }
    };
    // ---- one pipeline step on buffer `buf` = {K(next), V(cur)}:
    //   K-fragment reads -> [max / rare rescale of S(cur): VALU] -> wait -> QK^T MFMAs of the next tile
    //   V-fragment reads -> [exponentials of S(cur): VALU]       -> wait -> PV MFMAs of the current tile
    auto step = [&](int buf, int t_next, const f32x16 (&sc)[2], f32x16 (&sn)[2]) {
        V8 kf[2][KSTEPS];
        read_k(buf, kf);
        float mx = sc[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (!__all(mx <= m_run)) {  // exact: when no row's max moves, alpha == 1 and the rescale is the identity
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            if (KMASK) l_run *= alpha;
#pragma unroll
            for (int t = 0; t < DTILES; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        fence_lds();
        qk(kf, t_next, sn);
        // PV in two halves of two key-steps each (keeps the live V^T fragments + probabilities at 32 VGPRs)
        const float mc = m_run * c;
        float psum = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            V8 vf[DTILES][2];
            read_v(buf, half, vf);
            V8 pb[2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = sc[half][8 * k2 + e];
                    float pv;
                    if (ABL == 1) pv = sv * c - mc;
                    else if (ABL == 2) pv = sc[half][0];
                    else pv = __builtin_amdgcn_exp2f(sv * c - mc);
                    if (KMASK) psum += pv;   // !KMASK: the row sum comes out of the PV MFMA (V^T row dh is all ones)
                    pb[k2][e] = (S)pv;
                }
            fence_lds();
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int t = 0; t < DTILES; ++t) {
                    if (ABL == 3) { asm volatile("" :: "v"(vf[t][k2]), "v"(pb[k2])); }
                    else o[t] = T16<DT>::mfma32(vf[t][k2], pb[k2], o[t]);
                }
        }
        if (KMASK) l_run += psum;
    };

    // ---- software pipeline.  LDS buffer b = j & 1 holds the PAIR {K(j+1), V(j)}: the QK^T MFMAs of tile j+1 sit in
    // the same basic block as the exponentials and PV MFMAs of tile j, so the matrix pipe works while the VALU
    // exponentiates; the scores are double-buffered in registers (sA / sB).  Tile indices past the end are clamped
    // (the last QK^T is redundant) so the loop body has no data-dependent control flow besides the rescale.
    V8 kr[KIT], vr[VIT];
    f32x16 sA[2], sB[2];
    load_k(0, kr);
    store_k(1, kr);                       // K(0) parks in buffer 1's K area for the prologue
    load_k(1, kr);
    load_v(0, vr);
    __syncthreads();
    {
        V8 kf0[2][KSTEPS];
        read_k(1, kf0);
        fence_lds();
        qk(kf0, 0, sA);                   // S(0)
    }
    store_k(0, kr);                       // pair 0 = {K(1), V(0)}
    store_v(0, vr);
    __syncthreads();
    int j = 0;
    for (; j + 1 < ntiles; j += 2) {
        load_k(j + 2, kr);                // pair j+1 = {K(j+2), V(j+1)} -> buffer 1
        load_v(j + 1, vr);
        step(0, j + 1, sA, sB);           // softmax + PV of tile j, QK^T of tile j+1
        store_k(1, kr);
        store_v(1, vr);
        __syncthreads();
        load_k(j + 3, kr);                // pair j+2 = {K(j+3), V(j+2)} -> buffer 0
        load_v(j + 2, vr);
        step(1, min(j + 2, ntiles - 1), sB, sA);
        store_k(0, kr);
        store_v(0, vr);
        __syncthreads();
    }
    if (j < ntiles) step(0, ntiles - 1, sA, sB);   // odd tile count: the last tile's V sits in buffer 0

    // ---- epilogue: normalise and store out[b, q, h*dh + d]
    float l_tot;
    if (KMASK) {
        l_tot = l_run + __shfl_xor(l_run, 32);
    } else {
        // sum_k P[k, q] was accumulated by the PV MFMAs in output row d = dh (the all-ones row of V^T, primx_hip.h):
        // tile dh/32, register (rr&3) + 4*(rr>>3), half-wave (rr>>2)&1 with rr = dh % 32; rescaled together with O.
        const int rr = dh & 31;
        const float mine = o[DTILES - 1][0] * 0.f;  // placeholder to keep types simple
        float lsum = mine;
#pragma unroll
        for (int t = 0; t < DTILES; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t == (dh >> 5) && r == ((rr & 3) + 4 * (rr >> 3))) lsum = o[t][r];
        l_tot = __shfl(lsum, l31 + 32 * ((rr >> 2) & 1));
    }
    const float inv = 1.0f / l_tot;
    const int q = q0 + wave * 32 + l31;
    if (q < nq) {
        const int b = bh / H, h = bh - b * H;
        S* orow = out + ((int64_t)b * nq + q) * ((int64_t)H * dh) + (int64_t)h * dh;
#pragma unroll
        for (int t = 0; t < DTILES; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = t * 32 + 8 * g + 4 * hi;
                if (d < dh) {
                    V4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (S)(o[t][4 * g + e] * inv);
                    *reinterpret_cast<V4*>(orow + d) = v;
                }
            }
    }
}

// PRIMX_ATTN_STAGGER=<n>: start-up delay (n * 512 cycles) of odd workgroup generations; default tuned on MI355X
static const int g_attn_stagger = [] {
    const char* e = getenv("PRIMX_ATTN_STAGGER");
    return e ? atoi(e) : 0;
}();

static const int g_attn_abl = [] {
    const char* e = getenv("PRIMX_ATTN_ABL");
    return e ? atoi(e) : 0;
}();

template <int DT, int KSTEPS, int DTILES, int KMASK>
void launch_attn(const void* Qp, const void* Kp, const void* Vt, void* out, int B, int H, int nq, int nq_pad, int nkv,
                 int nkv_pad, int dh, float c, hipStream_t st) {
    using S = typename T16<DT>::S;
    dim3 grid(B * H, nq_pad / BQ);
#define PRIMX_ATTN_LAUNCH(A)                                                                                         \
    hipLaunchKernelGGL((attn_kernel<DT, KSTEPS, DTILES, KMASK, A>), grid, dim3(256), 0, st, (const S*)Qp, (const S*)Kp, \
                       (const S*)Vt, (S*)out, H, nq, nq_pad, nkv, nkv_pad, dh, c, g_attn_stagger)
    if (DT == PRIMX_F16 && KSTEPS == 5 && g_attn_abl != 0) {
        switch (g_attn_abl) {
            case 1: PRIMX_ATTN_LAUNCH(1); break;
            case 2: PRIMX_ATTN_LAUNCH(2); break;
            case 3: PRIMX_ATTN_LAUNCH(3); break;
            case 4: PRIMX_ATTN_LAUNCH(4); break;
            default: PRIMX_ATTN_LAUNCH(5); break;
        }
        return;
    }
    PRIMX_ATTN_LAUNCH(0);
#undef PRIMX_ATTN_LAUNCH
}

}  // namespace

extern "C" int primx_attention(const void* Qp, const void* Kp, const void* Vt, void* out, int B, int H, int nq,
                               int nq_pad, int nkv, int nkv_pad, int dh, float scale, int dtype, void* stream) {
    PRIMX_REQUIRE(Qp && Kp && Vt && out, "primx_attention: null pointer");
    PRIMX_REQUIRE(B > 0 && H > 0 && nq > 0 && nkv > 0, "primx_attention: empty problem");
    PRIMX_REQUIRE(nq_pad >= nq && nq_pad % BQ == 0, "primx_attention: nq_pad must be a multiple of 128 and >= nq");
    PRIMX_REQUIRE(nkv_pad >= nkv && nkv_pad % BKV == 0, "primx_attention: nkv_pad must be a multiple of 64 and >= nkv");
    PRIMX_REQUIRE(nq_pad / BQ <= 65535, "primx_attention: too many query tiles");
    const float c = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
    PRIMX_DISPATCH_16(dtype, "primx_attention", {
        if (dh == 72) launch_attn<DT, 5, 3, 0>(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, c, st);
        else if (dh == 64) launch_attn<DT, 4, 2, 1>(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, c, st);
        else if (dh == 32) launch_attn<DT, 2, 1, 1>(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, c, st);
        else {
            primx_set_error("primx_attention: unsupported head dim %d (supported: 32, 64, 72)", dh);
            return PRIMX_EINVAL;
        }
    });
    PRIMX_CHECK_LAUNCH("primx_attention");
    return PRIMX_OK;
}
