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
// KV tiles of 64 keys go global -> registers -> LDS with prefetch distance 2: two register sets, two LDS
// buffers, a branch-free loop body (tile indices past the end are clamped; only the LDS stores are
// predicated) so that hipcc keeps counted vmcnt waits - the v1 loop (one set, conditional loads) was
// compiled to s_waitcnt vmcnt(0) at the loop head.  One barrier per tile.  LDS rows are padded so
// that the 16-byte-slot stride is odd (K rows DP+8 halves, V^T rows 72 halves): ds_read_b128 conflict-free.
// Head dim 72 is zero-padded to DP = 80 for QK^T (5 k-steps) and to 96 output rows for PV (3 tiles):
// 22 MFMAs per 64 keys x 32 queries = 0.72 MFLOP issued for 0.59 MFLOP algorithmic (81.8 %).
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

template <int DT, int KSTEPS, int DTILES>
__global__ __launch_bounds__(256, 2) void attn_kernel(const typename T16<DT>::S* __restrict__ Qp,
                                                      const typename T16<DT>::S* __restrict__ Kp,
                                                      const typename T16<DT>::S* __restrict__ Vt,
                                                      typename T16<DT>::S* __restrict__ out, int H, int nq, int nq_pad,
                                                      int nkv, int nkv_pad, int dh, float c /* scale * log2(e) */) {
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
    auto load_tile = [&](int j, V8 (&kr)[KIT], V8 (&vr)[VIT]) {
        const S* kt = Kbase + (int64_t)j * BKV * DP;
        const S* vt = Vbase + j * BKV;
#pragma unroll
        for (int i = 0; i < KIT; ++i) kr[i] = ldg16<V8>(kt + k_goff[i]);
#pragma unroll
        for (int i = 0; i < VIT; ++i) vr[i] = ldg16<V8>(vt + v_goff[i]);
    };
    auto store_tile = [&](int buf, V8 (&kr)[KIT], V8 (&vr)[VIT]) {
        S* kb = smem + buf * BUF;
        S* vb = kb + KT;
#pragma unroll
        for (int i = 0; i < KIT; ++i)
            if (k_on[i]) *reinterpret_cast<V8*>(kb + k_loff[i]) = kr[i];
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

    // ---- one KV tile: S^T = K Q^T, online softmax, O^T += V^T P^T
    auto process = [&](int buf, int j) {
        const S* kb = smem + buf * BUF;
        const S* vb = kb + KT;
        f32x16 sc[2];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
            const S* krow = kb + (kt * 32 + l31) * KROW + hi * 8;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                V8 a = *reinterpret_cast<const V8*>(krow + s * 16);
                sc[kt] = T16<DT>::mfma32(a, qf[s], sc[kt]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if ((j + 1) * BKV > nkv) {  // mask the tail keys of the last tile
            const int kbase = j * BKV + 4 * hi;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= nkv) sc[kt][r] = -1e30f;
                }
        }
        // online softmax: per-lane scalars; one cross-half exchange of the max
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
            l_run *= alpha;
#pragma unroll
            for (int t = 0; t < DTILES; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        const float mc = m_run * c;
        float psum = 0.f;
        // key-step outer: the PV MFMAs of key-step ks issue as soon as ITS 8 probabilities are ready and run
        // while the VALU is still exponentiating key-steps ks+1.. (MFMA and VALU are separate pipes)
        const S* vrow = vb + l31 * VROW + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            V8 pb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = __builtin_amdgcn_exp2f(sc[ks >> 1][8 * (ks & 1) + e] * c - mc);
                psum += pv;
                pb[e] = (S)pv;
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < DTILES; ++t) {
                V8 a = *reinterpret_cast<const V8*>(vrow + t * 32 * VROW + ks * 16);
                o[t] = T16<DT>::mfma32(a, pb, o[t]);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        l_run += psum;
    };

    // ---- main loop: prefetch distance 2 (set A / set B), branch-free body
    const int ntiles = (nkv + BKV - 1) / BKV;
    V8 kA[KIT], vA[VIT], kB[KIT], vB[VIT];
    load_tile(0, kA, vA);
    load_tile(min(1, ntiles - 1), kB, vB);
    store_tile(0, kA, vA);
    __syncthreads();
    int j = 0;
    for (; j + 1 < ntiles; j += 2) {
        load_tile(min(j + 2, ntiles - 1), kA, vA);
        process(0, j);
        store_tile(1, kB, vB);
        __syncthreads();
        load_tile(min(j + 3, ntiles - 1), kB, vB);
        process(1, j + 1);
        store_tile(0, kA, vA);
        __syncthreads();
    }
    if (j < ntiles) process(0, j);

    // ---- epilogue: normalise and store out[b, q, h*dh + d]
    const float l_tot = l_run + __shfl_xor(l_run, 32);
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

template <int DT, int KSTEPS, int DTILES>
void launch_attn(const void* Qp, const void* Kp, const void* Vt, void* out, int B, int H, int nq, int nq_pad, int nkv,
                 int nkv_pad, int dh, float c, hipStream_t st) {
    using S = typename T16<DT>::S;
    dim3 grid(B * H, nq_pad / BQ);
    hipLaunchKernelGGL((attn_kernel<DT, KSTEPS, DTILES>), grid, dim3(256), 0, st, (const S*)Qp, (const S*)Kp,
                       (const S*)Vt, (S*)out, H, nq, nq_pad, nkv, nkv_pad, dh, c);
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
        if (dh == 72) launch_attn<DT, 5, 3>(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, c, st);
        else if (dh == 64) launch_attn<DT, 4, 2>(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, c, st);
        else if (dh == 32) launch_attn<DT, 2, 1>(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, c, st);
        else {
            primx_set_error("primx_attention: unsupported head dim %d (supported: 32, 64, 72)", dh);
            return PRIMX_EINVAL;
        }
    });
    PRIMX_CHECK_LAUNCH("primx_attention");
    return PRIMX_OK;
}
