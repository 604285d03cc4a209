// Flash-style attention forward for gfx950 on the pre-laid-out operands written by the projection
// GEMMs' epilogues (primx_linear_heads) - the replacement for xformers.ops.memory_efficient_attention.
//
// Geometry: workgroup = 8 waves = 256 query rows of one (batch, head), ONE workgroup per CU (two waves per SIMD); each
// wave owns 32 query rows for the whole kernel and keeps them in the LANE dimension of both MFMAs:
//
//   S^T[key, q]  = mfma_32x32x16( A = K tile rows (key-major, from LDS),  B = Q^T (registers) )
//   O^T[d,   q] += mfma_32x32x16( A = V^T tile rows (d-major, from LDS),  B = P^T (registers) )
//
// so lane (q = lane & 31) holds, for ITS query, 16 scores per 32-key sub-tile and 16 output features
// per 32-feature tile: running max / sum / rescale are per-lane scalars and the only cross-lane
// traffic per KV tile is one exchange of the tile max between the two half-waves (v_permlane32_swap).
// The accumulator register r of S^T holds key (r&3) + 8*(r>>2) + 4*hi of its 32-key sub-tile; the
// PRIMX_HEADS_VT layout stores V^T with the 4-key quads of every 16 keys in the order {0,2,1,3}, which
// makes the 8 keys a lane owns in one 16-key MFMA step a contiguous 16-byte LDS read - P never
// leaves registers and needs no permute.
//
// Staging is LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a 3-stage ring shared by all 8 waves:
// no staging VGPRs, no ds_write.  The GLOBAL layouts are designed for it: K rows are stored with the padded stride
// DP+8 halves (PRIMX_HEADS_KROWS), so a 64-key K tile is one contiguous block whose lane-linear LDS image has the odd
// 16-byte-slot row stride that makes ds_read_b128 conflict-free; V^T tiles (DP rows x 128 B) land in unpadded
// 128-byte LDS rows with the chunk index XOR-swizzled by ((row>>1)&7), applied on the per-lane SOURCE address.
// Stage j % 3 holds the PAIR {K(j+1), V(j)}: K runs one tile ahead of V (scores double-buffered in registers).
//
// Schedule ("ping-pong", MI355X_MICROARCH.md "Two waves per SIMD"): a step of a wave is a LIGHT segment L - DMA issue,
// all LDS fragment reads of the step's first MFMAs, row max / rare rescale - and a MATRIX segment M - 22 MFMAs with
// the exponentials in their shadow.  Waves 0-3 (group 0) and waves 4-7 (group 1) sit on the same four SIMDs and run
// the SAME code half a step apart, separated by workgroup barriers:
//        group 0:   L0 | M0 | L1 | M1 | L2 | ...
//        group 1:      | L0 | M0 | L1 | M1 | ...
// so each SIMD's matrix pipe always has one wave in M while its partner does the latency-bound work.  Measured before
// this schedule (profiles/r1_attn_ablation.txt): with both waves of a SIMD in phase, the non-matrix part of a step
// (~1500 of ~2900 cycles: LDS-read bursts of all 8 waves at once, LDS-DMA issue back-pressure, barrier skew) was
// fully exposed.
// Head dim 72 is zero-padded to DP = 80 for QK^T (5 k-steps) and to 96 output rows for PV (3 tiles):
// 22 MFMAs per 64 keys x 32 queries = 0.72 MFLOP issued for 0.59 MFLOP algorithmic (81.8 %).  The padding is put
// to work: column 72 of Q/K carries the key-padding mask and row 72 of V^T is all ones, so the MFMAs themselves
// deliver masked scores and the softmax denominator.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

constexpr int QPAD = 128;   // granularity of nq_pad (the Q operand buffers)
constexpr int NW = 8;       // waves per workgroup: all of them share every K / V^T tile the workgroup stages
constexpr int BQ = 32 * NW; // query rows per workgroup
constexpr int BKV = 64;     // keys per tile
constexpr int NSTAGE = 3;  // LDS ring depth (3..6 fit one workgroup per CU; deeper rings measured no better)
constexpr float RESCALE_THR = 8.0f;   // exp2-domain growth of the row max tolerated before O is rescaled
constexpr int WAITCNT_LGKM0 = 0xC07F;   // s_waitcnt simm16 on gfx9: vmcnt = 63 (no wait), expcnt = 7 (no wait), lgkmcnt = 0

template <typename V8>
__device__ __forceinline__ V8 ldg16(const void* ptr) {
    typedef __attribute__((address_space(1))) const V8 GV8;
    return *reinterpret_cast<GV8*>(reinterpret_cast<uintptr_t>(ptr));
}

// PROF = 1: phase profile - cycle stamps at the segment boundaries, summed per wave into g_attn_prof (PRIMX_ATTN_PROF=1).
// (The schedule variants that lost their same-box A/Bs - register staging, DMA issue / first exponentials / second V^T half in
// the other segment, deeper rings, static priorities, round-to-nearest pack, the "one wave group idle" probe - were build-time
// switches until round 4; their measurements are in DESIGN_LOG.md section 5.)
__device__ unsigned long long g_attn_prof[8];

template <int DT, int KSTEPS, int DTILES, int KMASK, int PROF = 0>
__global__ __launch_bounds__(64 * NW, 1) void attn_kernel(const typename T16<DT>::S* __restrict__ Qp,
                                                          const typename T16<DT>::S* __restrict__ Kp,
                                                          const typename T16<DT>::S* __restrict__ Vt,
                                                          typename T16<DT>::S* __restrict__ out, int H, int nq, int nq_pad,
                                                          int nkv, int nkv_pad, int dh, float c /* scale * log2(e) */,
                                                          const typename T16<DT>::S* __restrict__ Kb,
                                                          const typename T16<DT>::S* __restrict__ Vb, int b_from, int nkv_pad_b) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    using V4 = typename T16<DT>::V4;
    typedef __attribute__((address_space(1))) const void GV;
    typedef __attribute__((address_space(3))) void LV;
    constexpr int DP = 16 * KSTEPS;
    constexpr int KROW = DP + 8;               // K row stride in halves - in GLOBAL memory and in LDS
    constexpr int VR = 32 * DTILES;            // V^T rows an MFMA A-operand can touch
    constexpr int KT = BKV * KROW;             // halves in one K tile: whole 1 KiB DMA instructions
    constexpr int VT_ = VR * 64;               // halves in one V^T tile image (unpadded 128-byte rows)
    constexpr int BUF = KT + VT_;
    constexpr int NK = KT / 512;               // DMA wave-instructions per K tile   (11 / 9 / 5 for dh 72 / 64 / 32)
    constexpr int NV = DP / 8;                 // per V^T tile, 8 rows each          (10 / 8 / 4)
    constexpr int HW = NW / 2;                 // waves per group = per DMA role (group 0: K pieces, group 1: V^T pieces)
    constexpr int NSLOT = ((NK > NV ? NK : NV) + HW - 1) / HW;   // DMAs per wave per pair
    static_assert(KT % 512 == 0, "K tile must be whole DMA instructions");
    __shared__ __attribute__((aligned(16))) S smem[NSTAGE * BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / HW, rw = wave % HW;                                  // wave-uniform
    const int l31 = lane & 31, hi = lane >> 5;
    // Workgroup id -> (batch-head, query tile).  Ids go round-robin over the 8 XCDs, so with B*H % 8 == 0 all query tiles of a head share
    // an XCD's L2 whatever the order - but the ORDER decides whether its K / V^T (0.7 MB at 2048 keys) are still there for the next query
    // tile: with (batch, head) fastest an XCD's 32 CUs hold 32 different heads at a time (22 MB of operands through a 4 MB L2) and every
    // query tile re-fetches its head's keys from the fabric - 1577 MB per launch at batch 8 against 180 MB of operands
    // (profiles/r6_largeM_diagnosis.txt section 5).  Query tiles fastest INSIDE an XCD (round 6): the XCD's l-th workgroup is query
    // tile l % nqt of its head l / nqt - 32 CUs hold 32 / nqt heads.
    int bh = blockIdx.x, qt = blockIdx.y;
    if ((gridDim.x & 7) == 0) {
        const int id = blockIdx.x + gridDim.x * blockIdx.y, l = id >> 3, j = l / (int)gridDim.y;
        qt = l - j * (int)gridDim.y;
        bh = (id & 7) + 8 * j;
    }
    const int q0 = qt * BQ;
    // Batch entries >= b_from attend to nkv COPIES OF ONE key / value row (primx_attention_bcast: the unconditional half of
    // classifier-free guidance, whose conditioning tokens are one embedding expanded to the sequence length).  Their operands
    // hold that sequence once - tile 0 = 64 copies, tile 1 = the ragged last tile (nkv % 64 keys, the rest masked like any
    // pad rows) - in ONE entry shared by all of them; the walk over the nkv keys is unchanged, only the tile ADDRESS maps
    // j -> (j is the ragged last tile ? 1 : 0).  Same tile contents, same arithmetic, 1/11 of the bytes and L2-resident.
    const bool bcast = (bh / H) >= b_from;                                      // workgroup-uniform
    const int kvp = bcast ? nkv_pad_b : nkv_pad;
    const S* Kbase = bcast ? Kb + (int64_t)(bh % H) * kvp * KROW : Kp + (int64_t)bh * kvp * KROW;
    const S* Vbase = bcast ? Vb + (int64_t)(bh % H) * DP * kvp : Vt + (int64_t)bh * DP * kvp;

    // zero the V^T rows >= DP of every stage once (the DMA never writes them; they only feed discarded output rows)
    if (VR > DP) {
        for (int i = tid; i < (VR - DP) * 64; i += 64 * NW) {
#pragma unroll
            for (int stg = 0; stg < NSTAGE; ++stg) smem[stg * BUF + KT + DP * 64 + i] = (S)0.f;
        }
    }

    // Q^T B-operand fragments: lane (q = l31, hi) holds d = 16 s + 8 hi .. +7   (waves past nq_pad idle along on row nq_pad-1)
    // QCOL (dh + 3 <= DP, i.e. dh = 72): Q is PRE-SCALED by c = scale * log2(e) (one 16-bit rounding, 2^-11 relative, like the
    // rounding q already carries), so the MFMA delivers exp2-domain logits, and its spare padded columns dh+1 / dh+2 hold
    // -m_hi / -m_lo - the running row max split into two 16-bit halves - against constant ones in K (ops.alloc_heads): the
    // MFMA itself subtracts the max (exact products, fp32 accumulation) and a probability is ONE v_exp_f32 of an accumulator
    // register: no scale / subtract VALU work (16 v_pk_fma_f32 per wave and tile before).
    constexpr bool QCOL = !KMASK && (16 * KSTEPS - 3 >= 16 * (KSTEPS - 1) + 8);   // the three columns lie in the hi half of the last fragment
    // (loaded by load_q() BEHIND the prologue's DMA issue: with the Q loads first hipcc waited vmcnt(0) for them - and for the three K(0)
    // pieces issued after them - before it scaled Q, and only then issued pairs 0 and 1: two memory round trips in series in front
    // of the first MFMA of every launch)
    V8 qf[KSTEPS];
    auto load_q = [&]() {
        const S* qrow = Qp + ((int64_t)bh * nq_pad + min(q0 + wave * 32 + l31, nq_pad - 1)) * DP + hi * 8;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) qf[s] = ldg16<V8>(qrow + s * 16);
        if (QCOL) {
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[s][e] = (S)((float)qf[s][e] * c);
        }
    };
    const int dq = dh - 16 * (KSTEPS - 1) - 8 * hi;      // position of column dh in this lane's last fragment (hi lanes: 0 at dh = 72)
    auto set_q_cols = [&](float m) {                    // columns dh, dh+1, dh+2 of Q: 1, -m_hi, -m_lo
        const S mh = (S)(-m);
        const S ml = (S)(-m - (float)mh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e == dq) qf[KSTEPS - 1][e] = (S)1.0f;
            if (e == dq + 1) qf[KSTEPS - 1][e] = mh;
            if (e == dq + 2) qf[KSTEPS - 1][e] = ml;
        }
    };

    // ---- DMA issue.  Piece t of a pair: t < NK -> 1 KiB piece t of the (contiguous) K tile; else V^T rows
    // 8(t-NK) .. +7, lane (row = lane>>3, LDS chunk = lane&7) fetching the source chunk (lane&7) ^ ((row>>1)&7).  The
    // LDS image of a pair is contiguous (K tile, then V^T rows), so piece t always lands at stage + t KiB.
    // Roles are static: group 0 copies the K pieces, group 1 the V^T pieces, each wave a contiguous run; a wave whose
    // run is shorter than NSLOT re-issues its last piece (same bytes, harmless) and tile indices past the end are
    // clamped, so EVERY wave issues exactly NSLOT DMAs per pair with scalar-only address arithmetic - no branches, and
    // the vmcnt bookkeeping is static: "vmcnt(NSLOT)" = everything older than the newest pair has landed.
    const int ntiles = (nkv + BKV - 1) / BKV;
    const int n_pc = grp ? NV : NK, per = (n_pc + HW - 1) / HW;
    const int run_first = min(rw * per, n_pc - 1);
    const int run_len = max(min(per, n_pc - rw * per), 1);
    const int64_t piece_stride = grp ? (int64_t)8 * kvp : 512;                  // halves between consecutive pieces
    const int tile_stride = grp ? BKV : KT;
    const S* role_base = grp ? Vbase : Kbase;
    const int v_lrow = lane >> 3, v_lc = lane & 7;
    // per-lane source offset inside a piece; for V^T it depends on the parity of the 8-row group (swizzle term 4*tv & 7)
    const int lane_off0 = grp ? v_lrow * kvp + ((v_lc ^ ((v_lrow >> 1) & 7)) * 8) : lane * 8;
    const int lane_off1 = grp ? v_lrow * kvp + ((v_lc ^ ((4 + (v_lrow >> 1)) & 7)) * 8) : lane * 8;
    const bool ragged = (nkv & (BKV - 1)) != 0;
    auto tile_at = [&](int tile) {            // tile index inside this entry's operand buffers (uniform)
        const int tl = min(tile, ntiles - 1);
        return bcast ? ((ragged && tl == ntiles - 1) ? 1 : 0) : tl;
    };
    auto issue_run = [&](int tile, int stage) {   // this wave's pieces of K(tile) (group 0) / V(tile) (group 1)
        const S* tb = role_base + (int64_t)tile_at(tile) * tile_stride;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int pc = run_first + min(i, run_len - 1);
            const S* src = tb + pc * piece_stride + ((pc & 1) ? lane_off1 : lane_off0);
            __builtin_amdgcn_global_load_lds((GV*)(uintptr_t)src, (LV*)(smem + stage * BUF + (grp * NK + pc) * 512), 16, 0, 0);
        }
    };
    [[maybe_unused]] auto issue_pair = [&](int jp, int stage) { issue_run(jp + 1 - grp, stage); };   // {K(jp+1), V(jp)}

    f32x16 o[DTILES];
#pragma unroll
    for (int t = 0; t < DTILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = QCOL ? 0.f : -1e30f, l_run = 0.f;      // QCOL: the max the Q columns currently subtract (exp2 domain)
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    // ---- LDS fragment reads: batched, each batch consumed after ONE wait (builtin s_waitcnt: hipcc's own waitcnt
    // pass sees it and does not add a second, stricter wait in front of the consumers)
    auto read_k = [&](int stage, V8 (&kf)[2][KSTEPS]) {
        const S* kb = smem + stage * BUF + l31 * KROW + hi * 8;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) kf[kt][s] = *reinterpret_cast<const V8*>(kb + kt * 32 * KROW + s * 16);
    };
    auto read_v = [&](int stage, int half, V8 (&vf)[DTILES][2]) {   // key-steps 2*half, 2*half+1 of the tile
        const S* vb = smem + stage * BUF + KT + l31 * 64;
        const int sw = (l31 >> 1) & 7;                                // rows t*32 + l31: (row>>1)&7 does not depend on t
#pragma unroll
        for (int t = 0; t < DTILES; ++t)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
                vf[t][k2] = *reinterpret_cast<const V8*>(vb + t * 32 * 64 + (((4 * half + 2 * k2 + hi) ^ sw) * 8));
    };
    auto fence_lds = [&]() {
        __builtin_amdgcn_s_waitcnt(WAITCNT_LGKM0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // S^T(tile t) = K(t) Q^T.  Keys >= nkv: when the head dim has a spare padded column (dh < DP, e.g. 72 -> 80) they
    // are masked BY THE OPERANDS (Q[:, dh] = 1, K[pad rows, dh] = -30000, see primx_hip.h) and no code is needed here;
    // otherwise (KMASK) the scores are overwritten.
    auto qk = [&](const V8 (&kf)[2][KSTEPS], int t, f32x16 (&sc)[2]) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)   // the two 32-key chains alternate
                sc[kt] = T16<DT>::mfma32(kf[kt][s], qf[s], s == 0 ? zero16 : sc[kt]);  // shared zero C operand
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
    auto pv = [&](const V8 (&vf)[DTILES][2], const V8 (&pb)[2]) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int t = 0; t < DTILES; ++t) o[t] = T16<DT>::mfma32(vf[t][k2], pb[k2], o[t]);
    };
    // probabilities of one 32-key half of the tile (two 16-key MFMA steps): packed-fp32 FMA, v_exp, 16-bit pack
    auto probs = [&](const f32x16& sh, float mc, float& psum, V8 (&pb)[2]) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                if constexpr (QCOL) {   // the score already is (s * c - m): one exponential, round-toward-zero pack
                    const float p0 = __builtin_amdgcn_exp2f(sh[8 * k2 + e]), p1 = __builtin_amdgcn_exp2f(sh[8 * k2 + e + 1]);
                    if constexpr (DT == PRIMX_F16) {
                        const auto h2 = __builtin_amdgcn_cvt_pkrtz(p0, p1);
                        pb[k2][e] = h2[0];
                        pb[k2][e + 1] = h2[1];
                    } else {
                        const unsigned u = __builtin_amdgcn_perm(__float_as_uint(p1), __float_as_uint(p0), 0x07060302u);
                        typedef S S2 __attribute__((ext_vector_type(2)));
                        const S2 h2 = __builtin_bit_cast(S2, u);
                        pb[k2][e] = h2[0];
                        pb[k2][e + 1] = h2[1];
                    }
                } else {
                    const f32x2 sv = {sh[8 * k2 + e], sh[8 * k2 + e + 1]};
                    const f32x2 arg = __builtin_elementwise_fma(sv, (f32x2){c, c}, (f32x2){-mc, -mc});   // v_pk_fma_f32
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float pvv = __builtin_amdgcn_exp2f(arg[u]);
                        if (KMASK) psum += pvv;   // !KMASK: the row sum comes out of the PV MFMA (V^T row dh is all ones)
                        pb[k2][e + u] = (S)pvv;
                    }
                }
            }
    };

    // ---- segments.  WAITB: every DMA older than the newest pair has landed (each wave waits for its OWN pieces), all
    // LDS reads of the segment are home, then the workgroup barrier publishes both.
#define PRIMX_ATTN_WAITB_N(NF)                                                                                     \
    do {                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                     /* nothing of the next segment moves above ... */     \
        __builtin_amdgcn_s_waitcnt(((NF) & 15) | 0x70 | (((NF) >> 4) << 14)); /* vmcnt(NF) lgkmcnt(0) */           \
        asm volatile("s_barrier" ::: "memory");                                                                    \
        __builtin_amdgcn_sched_barrier(0);                     /* ... and nothing of this one sinks below */        \
    } while (0)
#define PRIMX_ATTN_WAITB() PRIMX_ATTN_WAITB_N(NFLY)
    constexpr int NFLY = NSLOT * (NSTAGE - 2);   // DMAs of the pairs newer than the one the next light segment reads
    // The barrier BEHIND a light segment.  The DMA of the next pair is issued in the MATRIX segment: nothing was issued since the previous matrix segment, and what that one
    // issued is read by the OTHER group's next light segment, which starts behind this very barrier - everything must
    // have landed (a first build waited vmcnt(NFLY) here too: stale tiles, NaNs on some launches).
    constexpr int NFLY_L = 0;
    unsigned long long pt = 0, pl = 0, pm = 0, pw = 0, pn = 0, pl_dma = 0, pl_rd = 0;
    auto stamp = [&](unsigned long long& acc) {
        if (PROF) {
            const unsigned long long now = __builtin_readcyclecounter();
            acc += now - pt;
            pt = now;
        }
    };
    [[maybe_unused]] V8 pb0_l[2];   // probabilities of the first 32 keys, formed in the light segment
    auto fold_max = [&](f32x16 (&sc)[2], bool first) {   // row max of a fresh score tile + the (rare) rescale
        float mx = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sc[0][r]), sc[1][r]);   // 16 x v_max3_f32
        {   // exchange with lane ^ 32: v_permlane32_swap, a VALU op (no LDS round trip behind the fragment reads)
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        // Deferred rescale (cdna_hip_programming.md T13): the running max is only raised - O rescaled - when some row's tile
        // max exceeds it by more than RESCALE_THR in the exp2 domain; until then the probabilities are formed against the
        // older max and are bounded by 2^RESCALE_THR = 256 (finite in fp16 / bf16, fp32 accumulation; the denominator is
        // accumulated from the same values, so the result is unchanged up to rounding).  With the exact rule ("whenever any
        // of the wave's 32 rows has a new max", RESCALE_THR = 0 up to the first tile) the branch fires on most tiles of random
        // data: P(some row's max moves at tile j) = 1 - (1 - 1/(j+1))^32 = 64 % even at j = 31.
        if constexpr (QCOL) {   // scores are relative to the max the Q columns carry; `first`: nothing is a max yet
            if (first || !__all(mx <= RESCALE_THR)) {
                // a REAL (scalar, uniform) branch: without this hipcc if-converts the body - ~90 VALU instructions issued with
                // an empty exec mask in every light segment (the first build of this path ran 65 instead of 54 us)
                asm volatile("" ::: "memory");
                const float delta = first ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run += delta;
                set_q_cols(m_run);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[kt][r] -= delta;   // the tile at hand was formed against the old max
#pragma unroll
                for (int t = 0; t < DTILES; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
            }
        } else if (!__all((mx - m_run) * c <= RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            if (KMASK) l_run *= alpha;
#pragma unroll
            for (int t = 0; t < DTILES; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
    };
    // LIGHT segment of step j on stage `st` = {K(j+1), V(j)}: DMA of pair j+2 into `st_refill` (last read, by either
    // group, before the barrier this segment started behind), fragment reads for QK^T(j+1) and the first half of PV(j),
    // row max of S(j) and the (rare) rescale of O.
    auto seg_light = [&](int st, int st_refill, int jp, f32x16 (&sc)[2], V8 (&kf)[2][KSTEPS], V8 (&vf0)[DTILES][2],
                         V8 (&vf1)[DTILES][2], bool first) {
        if (PROF) { __builtin_amdgcn_sched_barrier(0); stamp(pl_dma); }
        read_k(st, kf);
        read_v(st, 0, vf0);
        read_v(st, 1, vf1);
        if (PROF) { __builtin_amdgcn_sched_barrier(0); stamp(pl_rd); }
        fold_max(sc, first);
        {   // the exponentials of the tile's first 32 keys here: the matrix segment is at its issue bound, this one is not
            float psum0 = 0.f;
            probs(sc[0], m_run * c, psum0, pb0_l);
            if (KMASK) l_run += psum0;
        }
    };
    // MATRIX segment of step j: QK^T(j+1) with the exponentials of the first 32 keys of tile j in its shadow, PV of
    // those keys with the exponentials of the other 32 in its shadow, PV of the rest.  All operands of the first 16
    // MFMAs are in registers on entry; the second V^T half is read under the QK^T MFMAs.
    // (Moving the first exponentials into the light segment was measured: the matrix segment did not get shorter.)
    auto seg_matrix = [&](int st, int t_next, const f32x16 (&sc)[2], f32x16 (&sn)[2], const V8 (&kf)[2][KSTEPS],
                          const V8 (&vf0)[DTILES][2], V8 (&vf1)[DTILES][2], int jp, int st_refill) {
        __builtin_amdgcn_s_setprio(1);   // the matrix segment outranks its partner's light segment at the issue arbiter
        const float mc = m_run * c;
        float psum = 0.f;
        V8 pb0[2], pb1[2];
        // P of the first 32 keys is ready on entry: its 6 PV MFMAs are woven into the 10 QK^T MFMAs, so that no MFMA depends on
        // the one two slots before it (the two score chains alone sit exactly one MFMA latency apart: ~10 stall cycles each)
        pb0[0] = pb0_l[0];
        pb0[1] = pb0_l[1];
        {
            int pvi = 0;
#pragma unroll
            for (int s_ = 0; s_ < KSTEPS; ++s_) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) sn[kt] = T16<DT>::mfma32(kf[kt][s_], qf[s_], s_ == 0 ? zero16 : sn[kt]);
                if (pvi < 2 * DTILES) { o[pvi % DTILES] = T16<DT>::mfma32(vf0[pvi % DTILES][pvi / DTILES], pb0[pvi / DTILES], o[pvi % DTILES]); ++pvi; }
            }
#pragma unroll
            for (; pvi < 2 * DTILES; ++pvi) o[pvi % DTILES] = T16<DT>::mfma32(vf0[pvi % DTILES][pvi / DTILES], pb0[pvi / DTILES], o[pvi % DTILES]);
            if (KMASK && (t_next + 1) * BKV > nkv) {   // same key mask as qk() (head dims without a spare padded column)
                const int kbase = t_next * BKV + 4 * hi;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kbase + kt * 32 + (r & 3) + 8 * (r >> 2) >= nkv) sn[kt][r] = -1e30f;
            }
        }
        issue_pair(jp, st_refill);
        probs(sc[1], mc, psum, pb1);
        fence_lds();
        pv(vf1, pb1);
        if (KMASK) l_run += psum;
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: K(0) parks in the last stage, pairs 0 and 1 are issued; S(0) = K(0) Q^T
    if (!grp) issue_run(0, NSTAGE - 1);
#pragma unroll
    for (int pr = 0; pr < NSTAGE - 1; ++pr) issue_pair(pr, pr);
    load_q();
    if (QCOL) set_q_cols(0.f);
    f32x16 sA[2], sB[2];
    PRIMX_ATTN_WAITB();                   // K(0) and pair 0 landed
    {
        V8 kf0[2][KSTEPS];
        read_k(NSTAGE - 1, kf0);
        fence_lds();
        asm volatile("s_barrier" ::: "memory");   // group 0's L(0) refills this stage: everyone's K(0) reads are home first
        qk(kf0, 0, sA);                   // S(0)
    }
    if (PROF) pt = __builtin_readcyclecounter();
    // ---- main loop.  Group 0 runs  L B M B,  group 1 runs  B L B M  per step: the same number of barriers, group 1 half
    // a step late.  Ring safety (NSTAGE = 3): pair j+2 goes to the stage of pair j-1, whose K part was last read in
    // group 1's L(j-1) and whose V^T part in group 1's M(j-1) - both behind a barrier that precedes the issuing
    // segment (group 0 issues K pieces in its L(j), group 1 V^T pieces in its L(j)); pair j is complete before the
    // barrier in front of group 0's L(j) because every wave waits vmcnt(NSLOT) before EVERY barrier.
    int st = 0, st_free = NSTAGE - 1;
    V8 kf[2][KSTEPS], vf0[DTILES][2], vf1[DTILES][2];
    int j = 0;
    for (; j + 1 < ntiles; j += 2) {
        if (grp) { PRIMX_ATTN_WAITB(); stamp(pw); }
        seg_light(st, st_free, j + NSTAGE - 1, sA, kf, vf0, vf1, j == 0);
        stamp(pl);
        PRIMX_ATTN_WAITB_N(NFLY_L);
        stamp(pw);
        seg_matrix(st, j + 1, sA, sB, kf, vf0, vf1, j + NSTAGE - 1, st_free);      // softmax + PV of tile j, QK^T of tile j+1
        stamp(pm);
        if (!grp) { PRIMX_ATTN_WAITB(); stamp(pw); }
        st_free = st;
        st = (st == NSTAGE - 1) ? 0 : st + 1;
        if (grp) { PRIMX_ATTN_WAITB(); stamp(pw); }
        seg_light(st, st_free, j + NSTAGE, sB, kf, vf0, vf1, false);
        stamp(pl);
        PRIMX_ATTN_WAITB_N(NFLY_L);
        stamp(pw);
        seg_matrix(st, min(j + 2, ntiles - 1), sB, sA, kf, vf0, vf1, j + NSTAGE, st_free);
        stamp(pm);
        if (!grp) { PRIMX_ATTN_WAITB(); stamp(pw); }
        st_free = st;
        st = (st == NSTAGE - 1) ? 0 : st + 1;
        if (PROF) pn += 2;
    }
    if (j < ntiles) {                     // odd tile count (the DMAs are clamped and redundant: uniform vmcnt bookkeeping)
        if (grp) PRIMX_ATTN_WAITB();
        seg_light(st, st_free, j + NSTAGE - 1, sA, kf, vf0, vf1, j == 0);
        PRIMX_ATTN_WAITB_N(NFLY_L);
        seg_matrix(st, ntiles - 1, sA, sB, kf, vf0, vf1, j + NSTAGE - 1, st_free);
        if (!grp) PRIMX_ATTN_WAITB();
    }
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): drain the clamped tail DMAs before the workgroup retires
#undef PRIMX_ATTN_WAITB
#undef PRIMX_ATTN_WAITB_N
    if (PROF && lane == 0) {
        atomicAdd(&g_attn_prof[0], pl); atomicAdd(&g_attn_prof[1], pm); atomicAdd(&g_attn_prof[2], pw);
        atomicAdd(&g_attn_prof[3], pn); atomicAdd(&g_attn_prof[4], pl_dma); atomicAdd(&g_attn_prof[5], pl_rd);
    }

    // ---- epilogue: normalise and store out[b, q, h*dh + d]
    float l_tot;
    if (KMASK) {
        l_tot = l_run + __shfl_xor(l_run, 32);
    } else {
        // sum_k P[k, q] was accumulated by the PV MFMAs in output row d = dh (the all-ones row of V^T, primx_hip.h):
        // tile dh/32, register (rr&3) + 4*(rr>>3), half-wave (rr>>2)&1 with rr = dh % 32; rescaled together with O.
        const int rr = dh & 31;
        float lsum = 0.f;
#pragma unroll
        for (int t = 0; t < DTILES; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t == (dh >> 5) && r == ((rr & 3) + 4 * (rr >> 3))) lsum = o[t][r];
        l_tot = __shfl(lsum, l31 + 32 * ((rr >> 2) & 1));
    }
    const float inv = 1.0f / l_tot;
    const int q = q0 + wave * 32 + l31;
    const int b = bh / H, h = bh - b * H;
    // Row-major stores through LDS.  A lane owns one query row, so storing from the accumulators is 64 separate 8-byte
    // requests per instruction (9 instructions per wave at dh = 72, every wave at once at the end of the kernel); each wave
    // instead parks its 32 x dh tile in the idle ring (row stride DP + 8 halves) and walks it in 16-byte pieces, dh / 8
    // consecutive lanes per output row.
    if ((dh & 7) == 0) {
        constexpr int OST = DP + 8;
        static_assert(NW * 32 * OST <= NSTAGE * BUF, "output staging must fit the ring");
        __syncthreads();                                  // every wave has left the ring
        S* ost = smem + wave * (32 * OST);
#pragma unroll
        for (int t = 0; t < DTILES; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = t * 32 + 8 * g + 4 * hi;
                if (d < dh) {
                    V4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (S)(o[t][4 * g + e] * inv);
                    *reinterpret_cast<V4*>(ost + l31 * OST + d) = v;
                }
            }
        const int cpr = dh >> 3, npiece = 32 * cpr;       // 16-byte pieces per row / per wave tile
        for (int p = lane; p < npiece; p += 64) {
            const int row = p / cpr, c = p - row * cpr;
            const int qr = q0 + wave * 32 + row;
            if (qr < nq)
                *reinterpret_cast<V8*>(out + ((int64_t)b * nq + qr) * ((int64_t)H * dh) + (int64_t)h * dh + 8 * c) =
                    *reinterpret_cast<const V8*>(ost + row * OST + 8 * c);
        }
        return;
    }
    if (q < nq) {
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

// (Round 2 also built a ONE-wave-per-SIMD variant - 64 query rows per wave, O in the accumulator file, register-staged K / V^T,
// hand-placed VALU fillers between MFMAs - and measured it slower.  A single wave per SIMD issues one VALU instruction per
// ~4 cycles, half the rate two waves reach, and this kernel is bound by VALU issue (exp, pack, max), not by the matrix pipe.
// Removed in round 3; the measurements are in DESIGN_LOG.md section 5.)

// ---------------------------------------------------------------------------------------------------
// Small problems: nq, nkv <= 64 at dh = 32 - the VAE mid-block attention over the 64 voxels of a 4^3 primitive
// (models/vae3d_dib.py:168-186: 2048 primitives x 8 heads = 16,384 problems per sample).  The 8-wave kernel above gives one
// workgroup of 256 query rows to each of them: six of its eight waves idle, the operand buffers padded to 256 tokens, 198 us per
// decode at 43 TFLOP/s.  Here ONE WAVE owns a problem and nothing goes through LDS: the compact operand layouts (n_pad = 64) are
// already in MFMA fragment order - Q / K rows give the 16-byte operand pieces of S^T = K Q^T directly, the PRIMX_HEADS_VT
// layout's quad order makes a lane's eight keys of a PV step one 16-byte piece of a V^T row - so the wave loads 13 KB, runs
// 8 + 8 MFMAs with the softmax of its 64 query rows in registers (queries in the lane dimension: per-lane max / sum, one
// exchange with lane ^ 32) and stores 4 KB.  Memory-bound: 213 MB in, 67 MB out per decode.
template <int DT>
__global__ __launch_bounds__(256) void attn64_kernel(const typename T16<DT>::S* __restrict__ Qp, const typename T16<DT>::S* __restrict__ Kp,
                                                     const typename T16<DT>::S* __restrict__ Vt, typename T16<DT>::S* __restrict__ out,
                                                     int BH, int H, int nq, int nkv, float c /* scale * log2(e) */) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    using V4 = typename T16<DT>::V4;
    constexpr int NP = 64, DH = 32, KROW = DH + 8;
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bh >= BH) return;
    const S* q = Qp + (int64_t)bh * NP * DH;
    const S* k = Kp + (int64_t)bh * NP * KROW;
    const S* v = Vt + (int64_t)bh * DH * NP;
    V8 qf[2][2], kf[2][2], vf[2][2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[blk][ks] = ldg16<V8>(q + (blk * 32 + l31) * DH + ks * 16 + hi * 8);
            kf[blk][ks] = ldg16<V8>(k + (blk * 32 + l31) * KROW + ks * 16 + hi * 8);
            vf[blk][ks] = ldg16<V8>(v + l31 * NP + (4 * blk + 2 * ks + hi) * 8);   // keys of 16-group 2 blk + ks, this half-wave's quads
        }
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    const int b = bh / H, h = bh - b * H;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        f32x16 sc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            sc[kb] = T16<DT>::mfma32(kf[kb][0], qf[qb][0], zero16);
            sc[kb] = T16<DT>::mfma32(kf[kb][1], qf[qb][1], sc[kb]);
        }
        if (nkv < NP) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= nkv) sc[kb][r] = -1e30f;
        }
        float mx = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sc[0][r]), sc[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mc = mx * c;
        float psum = 0.f;
        V8 pb[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][8 * k2 + e], c, -mc));
                    psum += pv;
                    pb[kb][k2][e] = (S)pv;
                }
        psum += __shfl_xor(psum, 32);
        f32x16 o = zero16;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) o = T16<DT>::mfma32(vf[kb][k2], pb[kb][k2], o);
        const float inv = 1.0f / psum;
        const int qi = qb * 32 + l31;
        if (qi < nq) {
            S* orow = out + ((int64_t)b * nq + qi) * ((int64_t)H * DH) + (int64_t)h * DH;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                V4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (S)(o[4 * g + e] * inv);
                *reinterpret_cast<V4*>(orow + 8 * g + 4 * hi) = w;
            }
        }
    }
}

// PRIMX_ATTN_PROF=1: run the instrumented variant (dh 72, fp16) synchronously and print the per-segment cycle profile
static const int g_attn_prof_on = [] {
    const char* e = getenv("PRIMX_ATTN_PROF");
    return e ? atoi(e) : 0;
}();

template <int DT, int KSTEPS, int DTILES, int KMASK>
void launch_attn(const void* Qp, const void* Kp, const void* Vt, void* out, int B, int H, int nq, int nq_pad, int nkv,
                 int nkv_pad, int dh, float c, const void* Kb, const void* Vb, int b_from, int nkv_pad_b, hipStream_t st) {
    using S = typename T16<DT>::S;
    dim3 grid(B * H, (nq_pad + BQ - 1) / BQ);
#define PRIMX_ATTN_LAUNCH(P)                                                                                         \
    hipLaunchKernelGGL((attn_kernel<DT, KSTEPS, DTILES, KMASK, P>), grid, dim3(64 * NW), 0, st, (const S*)Qp,         \
                       (const S*)Kp, (const S*)Vt, (S*)out, H, nq, nq_pad, nkv, nkv_pad, dh, c, (const S*)Kb, (const S*)Vb,  \
                       b_from, nkv_pad_b)
    if constexpr (DT == PRIMX_F16 && KSTEPS == 5) {
        if (g_attn_prof_on) {
            unsigned long long z[8] = {0}, r[8];
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), z, sizeof(z));
            PRIMX_ATTN_LAUNCH(1);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_attn_prof), sizeof(r));
            const double n = r[3] ? (double)r[3] : 1.0;
            fprintf(stderr, "attn segment profile (cycles per step per wave, %llu wave-steps): light %.0f (DMA issue %.0f, LDS read "
                            "issue %.0f, max/rescale %.0f) | matrix %.0f | waits + barriers %.0f | total %.0f\n", r[3],
                    (r[0] + r[4] + r[5]) / n, r[4] / n, r[5] / n, r[0] / n, r[1] / n, r[2] / n, (r[0] + r[1] + r[2] + r[4] + r[5]) / n);
            return;
        }
    }
    PRIMX_ATTN_LAUNCH(0);
#undef PRIMX_ATTN_LAUNCH
}

}  // namespace

extern "C" int primx_attention(const void* Qp, const void* Kp, const void* Vt, void* out, int B, int H, int nq,
                               int nq_pad, int nkv, int nkv_pad, int dh, float scale, int dtype, void* stream) {
    return primx_attention_bcast(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, scale, nullptr, nullptr, B, 0, dtype, stream);
}

extern "C" int primx_attention_bcast(const void* Qp, const void* Kp, const void* Vt, void* out, int B, int H, int nq, int nq_pad,
                                     int nkv, int nkv_pad, int dh, float scale, const void* Kb, const void* Vb, int b_from,
                                     int nkv_pad_b, int dtype, void* stream) {
    PRIMX_REQUIRE(Qp && out && ((Kp && Vt) || b_from == 0), "primx_attention: null pointer");
    PRIMX_REQUIRE(B > 0 && H > 0 && nq > 0 && nkv > 0, "primx_attention: empty problem");
    PRIMX_REQUIRE(b_from >= 0 && b_from <= B, "primx_attention_bcast: b_from must lie in [0, B] (b_from=%d B=%d)", b_from, B);
    if (b_from < B) {
        const int need = ((nkv % 64) != 0 && nkv > 64) ? 128 : 64;
        PRIMX_REQUIRE(Kb && Vb, "primx_attention_bcast: null broadcast operands");
        PRIMX_REQUIRE(nkv >= 64 && nkv_pad_b >= need && nkv_pad_b % 64 == 0,
                      "primx_attention_bcast: needs nkv >= 64 and nkv_pad_b a multiple of 64 >= %d (nkv=%d nkv_pad_b=%d)", need, nkv, nkv_pad_b);
        PRIMX_REQUIRE(!(dh == 32 && nq <= 64 && nkv <= 64 && nq_pad == 64 && nkv_pad == 64),
                      "primx_attention_bcast: the one-wave 64-token kernel has no broadcast entries");
    }
    const bool small64 = dh == 32 && nq <= 64 && nkv <= 64 && nq_pad == 64 && nkv_pad == 64;   // one wave per problem (attn64_kernel)
    if (small64) {
        const float c64 = scale * 1.4426950408889634f;
        PRIMX_DISPATCH_16(dtype, "primx_attention", {
            using S = typename T16<DT>::S;
            hipLaunchKernelGGL((attn64_kernel<DT>), dim3((unsigned)((B * H + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const S*)Qp,
                               (const S*)Kp, (const S*)Vt, (S*)out, B * H, H, nq, nkv, c64);
        });
        PRIMX_CHECK_LAUNCH("primx_attention");
        return PRIMX_OK;
    }
    PRIMX_REQUIRE(nq_pad >= nq && nq_pad % QPAD == 0, "primx_attention: nq_pad must be a multiple of 128 and >= nq (or 64 for the 64-token, dh = 32 kernel)");
    PRIMX_REQUIRE(b_from == 0 || (nkv_pad >= nkv && nkv_pad % BKV == 0), "primx_attention: nkv_pad must be a multiple of 64 and >= nkv");
    PRIMX_REQUIRE(nq_pad / QPAD <= 65535, "primx_attention: too many query tiles");
    const float c = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
    PRIMX_DISPATCH_16(dtype, "primx_attention", {
        if (dh == 72) launch_attn<DT, 5, 3, 0>(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, c, Kb, Vb, b_from, nkv_pad_b, st);
        else if (dh == 64) launch_attn<DT, 4, 2, 1>(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, c, Kb, Vb, b_from, nkv_pad_b, st);
        else if (dh == 32) launch_attn<DT, 2, 1, 1>(Qp, Kp, Vt, out, B, H, nq, nq_pad, nkv, nkv_pad, dh, c, Kb, Vb, b_from, nkv_pad_b, st);
        else {
            primx_set_error("primx_attention: unsupported head dim %d (supported: 32, 64, 72)", dh);
            return PRIMX_EINVAL;
        }
    });
    PRIMX_CHECK_LAUNCH("primx_attention");
    return PRIMX_OK;
}
