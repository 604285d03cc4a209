// 16-bit MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (both operands K-contiguous: activations row-major,
// weights in nn.Linear (out,in) layout) with the DiT block's and the VAE decoder's epilogues fused.
//
// 256 threads = 4 waves; BK = 64; three tile shapes:
//   T144  : 128 x 144, waves 4x1, each wave 32 x 144 = 2 x 9 tiles of v_mfma_f32_16x16x32   (N % 144 == 0)
//           1152 = 8*144, 3456 = 24*144, 4608 = 32*144: with M = 4096 tokens (N_prim 2048 x CFG pair) the
//           DiT's GEMMs become exactly 256 / 768 / 1024 workgroups = 1 / 3 / 4 full waves of the 256 CUs.
//   Wide  : 128 x 128, waves 2x2, each wave 64 x 64 = 2 x 2 tiles of v_mfma_f32_32x32x16   (any N)
//   Narrow: 128 x  32, waves 4x1, each wave 32 x 32 = one 32x32x16 tile                   (N <= 32: VAE convs)
// Operand orientation: MFMA-A = activation rows (m), MFMA-B = weight rows (n), so the accumulator's
// lane index runs along n - the contiguous dimension of every destination - and each lane holds quads
// of 4 consecutive rows m.
//
// LDS: unpadded 128-byte rows (64 halves) with the 16-byte chunk index XOR-swizzled by ((row >> 1) & 7):
// conflict-free for the ds_read_b128 lane groups of BOTH MFMA shapes (checked exhaustively against the
// lane-group table of MI355X_MICROARCH.md) and for the 8-lane ds_write_b128 groups of the loader.
//
// Pipeline: register-staged, prefetch distance 2 - while k-tile t is multiplied out of LDS buffer t&1,
// the global loads of tiles t+1 AND t+2 are in flight in two register sets; tile t+1 is written to the
// other LDS buffer after the MFMAs (the compiler's counted vmcnt leaves tile t+2 in flight), one
// barrier per k-tile.  (v1 issued loads one tile ahead and measured ~1 us per k-tile - HBM/L2 latency
// bound at 300-450 TFLOP/s; see profiles/r1_kernel_trace_summary.txt.)
// Workgroup ids are remapped so each XCD (private 4 MiB L2) owns a contiguous run of tiles sharing A panels.
//
// GATHER = 1 turns the A loader into the implicit-GEMM gather of a 3x3x3 / stride 1 / pad 1 convolution
// over channels-last [P, S^3, Cin] activations: k = tap * Cin + ci, row m = (p, voxel); out-of-volume taps
// and any K tail read a 16-byte zero block instead of branching.
#include <stdio.h>
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "ln_row.h"

namespace {

// Global-address-space load of one 16-byte operand chunk.  Out-of-range chunks (K tail, conv padding
// taps) are handled by loading a VALID address and zeroing the VALUE: selecting between pointers of
// different provenance made hipcc emit flat_load (counted on vmcnt AND lgkmcnt, not partially waitable),
// which serialised every LDS wait behind the global prefetches (profiles/r1_gemm_pmc.txt).
template <typename V8, typename S>
__device__ __forceinline__ V8 ldg16(const S* ptr, bool keep) {
    typedef __attribute__((address_space(1))) const V8 GV8;
    V8 v = *reinterpret_cast<GV8*>(reinterpret_cast<uintptr_t>(ptr));
    if (!keep) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (S)0.f;
    }
    return v;
}

constexpr int BK = 64;

// (EPI_GATE_RESIDUAL_LN: gemm144l_dma_kernel only - the gate-residual epilogue followed by the LayerNorm + modulate of the row block)
// The *_FOLD epilogues are the LayerNorm fold (see "LayerNorm fold" below): EPI_GATE_RESIDUAL_FOLD = the PRODUCER of a folded
// LayerNorm site (gemm144l only), EPI_HEADS_FOLD / EPI_LINEAR_FOLD = its CONSUMERS (gemm144l, gemm288q),
// EPI_F32OUT = fp32 rows out of 16-bit operands (the per-timestep u / v vectors of the fold; generic kernel only).
enum { EPI_LINEAR = 0, EPI_GATE_RESIDUAL = 1, EPI_HEADS = 2, EPI_RES = 3, EPI_CONVT = 4, EPI_GATE_RESIDUAL_LN = 5,
       EPI_GATE_RESIDUAL_FOLD = 6, EPI_HEADS_FOLD = 7, EPI_LINEAR_FOLD = 8, EPI_F32OUT = 9 };

template <int DT>
struct GemmArgs {
    using S = typename T16<DT>::S;
    const S* A;
    const S* W;
    const S* bias;  // may be null
    int M, N, K;
    // EPI_LINEAR / EPI_RES / EPI_CONVT
    S* out;
    int act;
    float out_scale;
    // EPI_GATE_RESIDUAL
    const S* gate;
    int64_t gate_stride;
    float* x;
    int rows_per_batch;
    // EPI_HEADS
    int heads, dh, DP, n_pad, n_seg;
    int kind[3];
    int xcd_gm;  // 256x288 kernel: workgroups of one XCD form an (mt / xcd_gm) x (nt / (8 / xcd_gm)) block of tiles (0 / 8: whole tile rows)
    int prof;  // PRIMX_GEMM_PROF=1: per-workgroup timeline stamps into g_gemm_prof
    S* dst[3];
    float scale0;
    int64_t rep_stride[3];  // the n_seg column groups repeat; repetition r writes at dst[s] + r * rep_stride[s]
    // EPI_RES: out = cast16((acc + bias + res) * out_scale); res may be null
    const S* res;
    // conv gather (GATHER) and EPI_CONVT geometry
    int S3;        // grid edge S (volume S^3)
    int cin_log2;  // log2(Cin)
    int cout;      // EPI_CONVT: N = 8 * cout
    // the entry points' `prefetch` range: pf_lines 128-byte lines from pf_ptr that the compute waves of the loader-wave kernels touch once
    const char* pf_ptr;
    int64_t pf_lines;
    // primx_linear_gate_residual_ln: LayerNorm + modulate of the updated rows (ln_out = null: plain gate-residual).  `sync`: two
    // zero-initialised words per 128-row block (arrivals, departures); ln_light: same-XCD fences (see gemm144l_dma_kernel)
    const S* ln_shift;
    const S* ln_scale;
    int64_t ln_mod_stride;
    S* ln_out;
    float ln_eps;
    unsigned* sync;
    int ln_light;
    // LayerNorm fold.  Producer (EPI_GATE_RESIDUAL_FOLD): ln_scale / ln_mod_stride = the NEXT LayerNorm's scale vectors, ln_out = the
    // 16-bit operand cast16((x - c) rho_p (1 + scale)), fold_c = [M][2] per-row (centre c, scale rho_p) (read), fold_part =
    // [M][N / 144][2] partial sums of (x - c), (x - c)^2 per column tile (written).  Consumers: fold_part / fold_parts (read), fold_u /
    // fold_v = fp32 per-column vectors, fold_c = the (c, rho_p) the producer used (read), fold_c_out = [M][2] (written by the column
    // tile 0 workgroups: this site's (mean, rstd) = the next producer's (c, rho_p); never the array fold_c points at), fold_eps.
    float* fold_part;
    int fold_parts;
    const float* fold_u;
    const float* fold_v;
    const float* fold_c;
    float* fold_c_out;
    float fold_eps;
};

// Weight prefetch carried by a GEMM launch (the `prefetch` / `prefetch_bytes` arguments of primx_linear, primx_linear_heads,
// primx_linear_gate_residual[_ln]).  In the loader-wave kernels the compute waves never use their vector-memory queue inside the
// k-loop, so each of them can request one dword per 128-byte line of ANOTHER GEMM's weights in front of the loop - one or two
// instructions per wave - and the lines travel HBM -> Infinity Cache while the loop runs from L2, when the fabric is otherwise
// idle.  (Carried by the LayerNorm launches instead - primx_layernorm_modulate's pf0 / pf1 - the same bytes compete with the
// LayerNorm's own stream: 6.8 -> 8.3 us per launch.)  Kernels without loader waves ignore the range.
template <int DT>
static int set_prefetch(GemmArgs<DT>& a, const void* ptr, int64_t bytes, const char* name) {
    PRIMX_REQUIRE((ptr != nullptr) == (bytes > 0) && bytes >= 0, "%s: the prefetch range is (pointer, bytes > 0) or (NULL, 0)", name);
    a.pf_ptr = (const char*)ptr;
    a.pf_lines = bytes >= 4 ? (bytes - 4) / 128 + 1 : 0;   // one dword per line, every dword inside [ptr, ptr + bytes)
    return PRIMX_OK;
}

typedef unsigned int pf_u32x2 __attribute__((ext_vector_type(2)));
template <int DT>
__device__ __forceinline__ pf_u32x2 gemm_prefetch_lines(const GemmArgs<DT>& p, int wave, int lane) {
    // (the two values are handed back untouched - any arithmetic on them here would make hipcc wait for the loads on the spot - and
    // are "used" by an empty asm behind the k-loop, where the compiler's own vmcnt wait for them costs nothing)
    pf_u32x2 v = {0u, 0u};
    if (p.pf_lines > 0) {                                  // (uniform)
        const int64_t l0 = ((int64_t)blockIdx.x * 8 + wave) * 64 + lane, stride = (int64_t)gridDim.x * 512;
        // two independent requests per lane cover 1024 lines per workgroup: 33.5 MB with 256 workgroups; longer ranges are cut
        if (l0 < p.pf_lines) v[0] = *reinterpret_cast<const unsigned*>(p.pf_ptr + l0 * 128);
        if (l0 + stride < p.pf_lines) v[1] = *reinterpret_cast<const unsigned*>(p.pf_ptr + (l0 + stride) * 128);
    }
    return v;
}

// The same range from a kernel WITHOUT loader waves (gemm288q_dma_kernel, round 6): its waves wait on their vector-memory queue inside the
// k-loop, so the requests go out BEHIND the loop - in front of an epilogue that loads nothing - from the LAST round of workgroups (the
// last min(grid, 256) ids: a multi-round launch streams several times the Infinity Cache's size between its first round and its end).
// Measured why it matters (profiles/r6_fc1_onepass.txt): fc1 at T = 4096 on this kernel without the carry was 5.5 us faster than the
// two-pass kernel and made the fc2 launch behind it 8 us slower - fc2's 10.6 MB of weights then came from HBM.
template <int DT>
__device__ __forceinline__ pf_u32x2 gemm_prefetch_lines_tail(const GemmArgs<DT>& p, int wave, int lane) {
    pf_u32x2 v = {0u, 0u};
    if (p.pf_lines > 0) {                                  // (uniform)
        const int last = min((int)gridDim.x, 256), b = (int)blockIdx.x - ((int)gridDim.x - last);
        if (b >= 0) {
            const int64_t l0 = ((int64_t)b * 8 + wave) * 64 + lane, stride = (int64_t)last * 512;
            if (l0 < p.pf_lines) v[0] = *reinterpret_cast<const unsigned*>(p.pf_ptr + l0 * 128);
            if (l0 + stride < p.pf_lines) v[1] = *reinterpret_cast<const unsigned*>(p.pf_ptr + (l0 + stride) * 128);
        }
    }
    return v;
}

// Kernel arguments of the LDS-DMA kernels.  A struct passed by value lives in the kernarg segment and every wave starts with an
// s_load of it - a scalar-cache miss, since the packet processor has just written that memory - before it can form its first DMA
// address.  The fields the prologue needs come as LEADING SCALAR arguments instead, which the build preloads into SGPRs at wave
// launch (csrc/build.py: -mllvm -amdgpu-kernarg-preload-count=16; aggregates are not preloadable): -0.25 us per launch measured on
// the LayerNorm kernel, whose arguments are scalars anyway (rocprofv3, 3910 launches: 8.35 -> 8.10 us).  The rest of the struct is
// loaded behind the first DMAs.
#define PRIMX_GEMM_PARAMS(DT)                                                                                                \
    const typename T16<DT>::S *pl_A, const typename T16<DT>::S *pl_W, int pl_M, int pl_N, int pl_K, int pl_xcd_gm, int pl_prof, \
        const GemmArgs<DT> pl_rest
// (no local copy of the struct with the scalars patched in: the dynamically indexed members - kind[], dst[], rep_stride[] - would
// turn it into a scratch array; the kernels name the leading arguments directly and read everything else through `p`)
#define PRIMX_GEMM_ARGS(DT) const GemmArgs<DT>& p = pl_rest
#define PRIMX_GEMM_PASS(x) (x).A, (x).W, (x).M, (x).N, (x).K, (x).xcd_gm, (x).prof, (x)

// Activation of the EPI_LINEAR epilogue.  `p.act` is uniform, but both branches are pure arithmetic, so hipcc if-converts
// them: every element then paid for the tanh form AND the erf polynomial (~40 VALU instructions instead of ~12; the fc1
// epilogue was 31k of the kernel's 91k cycles, PRIMX_GEMM_PROF).  The empty volatile asm makes each side
// non-speculatable, which forces real (scalar, uniform) branches.
#define PRIMX_APPLY_ACT(y)                                   \
    do {                                                     \
        if (p.act == PRIMX_ACT_GELU_TANH) {                  \
            asm volatile("" ::: "memory");                   \
            y = rnd16<DT>(gelu_tanh_f(y));                   \
        } else if (p.act == PRIMX_ACT_GELU_ERF) {            \
            asm volatile("" ::: "memory");                   \
            y = rnd16<DT>(gelu_erf_f(y));                    \
        }                                                    \
    } while (0)

// Output stores of the epilogues.  PRIMX_STORE_POLICY (build-time experiment): 0 = plain (write-back: the lines stay dirty in
// the XCD's L2 until evicted or until the end-of-kernel release), 1 = non-temporal (`nt`), 2 = write-through (`sc0 sc1`,
// MI355X_MICROARCH.md "stores of each flavour": the line leaves L2 with the store).
#ifndef PRIMX_STORE_POLICY
#define PRIMX_STORE_POLICY 0
#endif
// PRIMX_LOADER_PRIO / PRIMX_COMPUTE_PRIO (build-time experiments): s_setprio of the loader / compute waves of the loader-wave kernels
// (0 = the default priority).  Measured in round 5 (tools/gpu/r5_prio.sh, same box, configs[1] step): loaders at 1 or 3, compute
// waves at 1: 8.61 - 8.62 ms against 8.60 - the issue arbiter is not what these kernels wait for.
#ifndef PRIMX_LOADER_PRIO
#define PRIMX_LOADER_PRIO 0
#endif
#ifndef PRIMX_COMPUTE_PRIO   // the same for their compute waves
#define PRIMX_COMPUTE_PRIO 0
#endif
template <typename T>
__device__ __forceinline__ void out_store(T* ptr, const T v) {
#if PRIMX_STORE_POLICY == 1
    __builtin_nontemporal_store(v, ptr);
#elif PRIMX_STORE_POLICY == 2
    if constexpr (sizeof(T) == 8) {
        const u32x2 w = __builtin_bit_cast(u32x2, v);
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(w) : "memory");
    } else if constexpr (sizeof(T) == 16) {
        const u32x4 w = __builtin_bit_cast(u32x4, v);
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(w) : "memory");
    } else {
        *ptr = v;
    }
#else
    *ptr = v;
#endif
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "contiguous chunk per XCD" remap (cdna_hip_programming.md T1)
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// 2-D variant: the 8 XCDs (private L2s) tile the grid of output tiles gm x (8 / gm); inside its block an XCD walks n fastest.
// With whole tile rows per XCD (gm = 8) every XCD streams the WHOLE weight matrix through its L2: fc1 at T = 4096 fetched
// 93 MB per launch for 20 MB of operands (PMC: A 9.4 + 8 x W 10.6); a 2 x 4 / 4 x 2 split replicates both operands a
// little instead of one of them eightfold.  Host picks gm (launch144_dma); needs mt % gm == 0 and nt % (8 / gm) == 0.
// Inside its bm x bn block an XCD takes the tiles in the order its 32 CUs should hold them TOGETHER: sub-blocks of sr x sc tiles
// (sr x sc ~ 32, both ~ 6), m fastest inside a sub-block, sub-blocks down a column strip, strips left to right.  A round of
// workgroups then streams sr + sc operand panels through the 4 MB L2 instead of 1 - 2 + bn with the row-major walk (fc1 at
// T = 32768: 2 + 16 panels of 0.6 MB per round, i.e. the whole weight matrix from the Infinity Cache in every round; the batched
// K / V projection: 1 + 28), and the workgroups of a round move through k together, so a panel's slice is fetched once per round.
// `gm_packed` = gm | sr << 8 | sc << 16 (xcd_pack, host): the sub-block shape is found once per launch, not once per workgroup.
__device__ __forceinline__ void xcd_tile2d(int bid, int mt, int nt, int gm_packed, int& mi, int& ni) {
    const int gm = gm_packed & 255, sr = (gm_packed >> 8) & 255, sc = gm_packed >> 16;
    const int gn = 8 / gm, bm = mt / gm, bn = nt / gn;
    const int x = bid & 7, local = bid >> 3;
    const int per_strip = bm * sc;
    const int st = local / per_strip, rem = local - st * per_strip;
    const int w = min(sc, bn - st * sc);                // (the last strip may be narrower)
    const int g = rem / (sr * w), rem2 = rem - g * (sr * w);
    const int c = rem2 / sr, r = rem2 - c * sr;
    mi = (x / gn) * bm + g * sr + r;
    ni = (x - (x / gn) * gn) * bn + st * sc + c;
}

// byte-free LDS addressing in halves: row-major 64-half rows, 16-byte chunks XOR-swizzled
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

template <int MF>
struct AccT;
template <>
struct AccT<32> {
    using T = f32x16;
};
template <>
struct AccT<16> {
    using T = f32x4;
};

// Per-column constants of the epilogue (computed once per lane per n-tile).
struct ColInfo {
    int n;
    bool ok;
    float bias;
    int seg, hh, dd, rep;
};

template <int DT, int EPI>
__device__ __forceinline__ ColInfo make_col(const GemmArgs<DT>& p, int n) {
    ColInfo c;
    c.n = n;
    c.ok = n < p.N;
    c.seg = c.hh = c.dd = c.rep = 0;
    if (EPI == EPI_HEADS && c.ok) {
        const int per = p.heads * p.dh;
        const int seg_all = n / per;
        c.rep = seg_all / p.n_seg;
        c.seg = seg_all - c.rep * p.n_seg;
        const int w = n - seg_all * per;
        c.hh = w / p.dh;
        c.dd = w - c.hh * p.dh;
    }
    if (EPI == EPI_CONVT && c.ok) {
        c.seg = n / p.cout;
        c.dd = n - c.seg * p.cout;
    }
    c.bias = (p.bias && c.ok) ? (float)p.bias[EPI == EPI_CONVT ? c.dd : n] : 0.f;
    return c;
}

// One accumulator quad: rows mq .. mq+3 (consecutive m), column c.n.
template <int DT, int EPI>
__device__ __forceinline__ void epilogue_quad(const GemmArgs<DT>& p, const ColInfo& c, int mq, const float (&a)[4]) {
    using S = typename T16<DT>::S;
    using V4 = typename T16<DT>::V4;
    if (!c.ok) return;
    const int n = c.n;
    if (EPI == EPI_LINEAR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mq + j;
            if (m >= p.M) continue;
            float y = rnd16<DT>(a[j] + c.bias);
            PRIMX_APPLY_ACT(y);
            if (p.out_scale != 1.0f) y = rnd16<DT>(p.out_scale * y);
            p.out[(int64_t)m * p.N + n] = (S)y;
        }
    } else if (EPI == EPI_RES) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mq + j;
            if (m >= p.M) continue;
            float y = a[j] + c.bias;
            if (p.res) y += (float)p.res[(int64_t)m * p.N + n];
            p.out[(int64_t)m * p.N + n] = (S)(y * p.out_scale);
        }
    } else if (EPI == EPI_CONVT) {
        // row m = (prim, z, y, x) on the S^3 grid; column = (tap dz,dy,dx ; co) -> voxel (2z+dz, ..)
        const int Sg = p.S3, V = Sg * Sg * Sg, S2 = 2 * Sg;
        const int dz = c.seg >> 2, dy = (c.seg >> 1) & 1, dx = c.seg & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mq + j;
            if (m >= p.M) continue;
            const int pp = m / V, v = m - pp * V;
            const int z = v / (Sg * Sg), y = (v / Sg) % Sg, x = v % Sg;
            const int64_t ov = ((int64_t)(2 * z + dz) * S2 + (2 * y + dy)) * S2 + (2 * x + dx);
            p.out[((int64_t)pp * 8 * V + ov) * p.cout + c.dd] = (S)(a[j] + c.bias);
        }
    } else if (EPI == EPI_GATE_RESIDUAL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mq + j;
            if (m >= p.M) continue;
            const int b = m / p.rows_per_batch;
            const float gt = (float)p.gate[(int64_t)b * p.gate_stride + n];
            const float v = rnd16<DT>(a[j] + c.bias);
            float* xp = p.x + (int64_t)m * p.N + n;
            *xp = *xp + rnd16<DT>(gt * v);
        }
    } else if (EPI == EPI_F32OUT) {
        // fp32 rows (p.x) of 16-bit operands; the bias joins the rows from p.rows_per_batch on (the fold's v rows, not its u rows)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mq + j;
            if (m >= p.M) continue;
            p.x[(int64_t)m * p.N + n] = a[j] + (m >= p.rows_per_batch ? c.bias : 0.f);
        }
    } else {  // EPI_HEADS
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rnd16<DT>(a[j] + c.bias);
        S* dst = p.dst[0];
        int kind = p.kind[0];
        int64_t rstride = p.rep_stride[0];
        if (c.seg == 1) { dst = p.dst[1]; kind = p.kind[1]; rstride = p.rep_stride[1]; }
        if (c.seg == 2) { dst = p.dst[2]; kind = p.kind[2]; rstride = p.rep_stride[2]; }
        dst += c.rep * rstride;
        if (c.seg == 0 && p.scale0 != 1.0f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = rnd16<DT>(p.scale0 * v[j]);
        }
        const int b0 = mq / p.rows_per_batch, tok0 = mq - b0 * p.rows_per_batch;
        if (kind == PRIMX_HEADS_VT && mq + 3 < p.M && tok0 + 3 < p.rows_per_batch && (tok0 & 1) == 0) {
            // 4 consecutive tokens of one batch entry.  tok0 % 4 == 0: one contiguous quad of the VT layout (8-byte store);
            // tok0 % 4 == 2 (rows_per_batch % 4 == 2, e.g. 1370 conditioning tokens, second batch entry): the second half
            // of one quad + the first half of the next (two 4-byte stores) - never 2-byte scalars.
            typedef S V2 __attribute__((ext_vector_type(2)));
            S* row = dst + (((int64_t)b0 * p.heads + c.hh) * p.DP + c.dd) * p.n_pad;
            if ((tok0 & 3) == 0) {
                V4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (S)v[j];
                *reinterpret_cast<V4*>(row + vt_key_pos(tok0)) = o;
            } else {
                *reinterpret_cast<V2*>(row + vt_key_pos(tok0)) = V2{(S)v[0], (S)v[1]};
                *reinterpret_cast<V2*>(row + vt_key_pos(tok0 + 2)) = V2{(S)v[2], (S)v[3]};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = mq + j;
                if (m >= p.M) continue;
                const int b = m / p.rows_per_batch, tok = m - b * p.rows_per_batch;
                const int64_t head = (int64_t)b * p.heads + c.hh;
                if (kind != PRIMX_HEADS_VT) dst[(head * p.n_pad + tok) * heads_row_stride(kind, p.DP) + c.dd] = (S)v[j];
                else dst[(head * p.DP + c.dd) * p.n_pad + vt_key_pos(tok)] = (S)v[j];
            }
        }
    }
}

// Row-major epilogue unit: ONE row m, FOUR consecutive columns n..n+3 (n % 4 == 0, all inside one head / one
// segment).  Every access is 8 bytes (16-bit destinations) or 16 bytes (the fp32 residual stream), and the 36 lanes
// covering a 144-column tile row touch 288 / 576 contiguous bytes - the quad-per-lane form touches 32 / 64.
// `bv`: the four bias values of columns n..n+3, loaded by the caller BEFORE its store loop (a load issued between
// the stores cannot be hoisted by the compiler - the output may alias it - and each one then costs a full memory
// round trip: 9 x ~775 cycles per tile, PRIMX_GEMM_PROF).
// EPI_LINEAR arithmetic of four consecutive columns: bias, rounding, activation, scale - each rounded to the 16-bit
// type like the separate autocast ops of the reference.
// rnd16 -> GELU-tanh -> rnd16 of four values in packed fp32 arithmetic (gelu_tanh4): the Linear + GELU epilogue when no scale follows
template <int DT>
__device__ __forceinline__ typename T16<DT>::V4 gelu_out4(const f32x4 y) {
    using V4 = typename T16<DT>::V4;
    return __builtin_convertvector(gelu_tanh4(__builtin_convertvector(__builtin_convertvector(y, V4), f32x4)), V4);
}

// GELU = true: the caller has established (one uniform branch per tile, not per quad) that p.act == PRIMX_ACT_GELU_TANH and
// p.out_scale == 1, and passes zeros in `bv` when there is no bias.
template <bool B>
struct BoolC {
    static constexpr bool value = B;
};
template <int DT, bool GELU = false>
__device__ __forceinline__ typename T16<DT>::V4 linear_out4(const GemmArgs<DT>& p, const f32x4 a,
                                                            const typename T16<DT>::V4 bv) {
    using S = typename T16<DT>::S;
    if constexpr (GELU) return gelu_out4<DT>(a + __builtin_convertvector(bv, f32x4));
    typename T16<DT>::V4 o;
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = rnd16<DT>(a[j] + (p.bias ? (float)bv[j] : 0.f));
    // The activation's own rounding is folded into the final conversion when no scale follows (rounding twice to the
    // same type is the identity), and the scale is a real uniform branch: if-converted it cost every element a
    // multiply, two conversions and a select on top of the ~13 issue slots of the activation itself (the fc1 epilogue
    // is bound by exactly this arithmetic: DESIGN_LOG.md section 4).
    if (p.act == PRIMX_ACT_GELU_TANH) {          // one uniform branch per four elements (see PRIMX_APPLY_ACT)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(y[j]);
    } else if (p.act == PRIMX_ACT_GELU_ERF) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = gelu_erf_f(y[j]);
    }
    if (p.out_scale != 1.0f) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (S)(p.out_scale * rnd16<DT>(y[j]));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (S)y[j];
    }
    return o;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm fold (primx_linear_gate_residual_fold -> primx_linear_heads_fold / primx_linear_fold).  Every LayerNorm + modulate of a
// DiT block sits between a gated residual add and a Linear (dit_crossattn.py:55-57).  With the row statistics mu, rho of the fp32
// residual stream x, m = cast16(1 + scale), ANY per-row centre c and ANY per-row scale rho_p > 0:
//     reference:  y = cast16( cast16( (x - mu) rho m + shift ) W^T + b )
//     folded:     y = cast16( (rho / rho_p) cast16((x - c) rho_p m) W^T  -  rho (mu - c) u  +  v ),   u = m W^T,  v = shift W^T + b  (fp32, per column)
// The PRODUCER (the gate-residual GEMM, EPI_GATE_RESIDUAL_FOLD) stores a16 = cast16((x - c) rho_p m) next to x and the partial
// sums of (x - c), (x - c)^2 of its 144 columns; the CONSUMER (to_q / qkv / fc1, EPI_HEADS_FOLD / EPI_LINEAR_FOLD) multiplies a16,
// finishes mu' = mean(x - c) and rho from the partials and applies them with u, v in its epilogue: no LayerNorm kernel and no
// second pass over the fp32 rows.  u, v depend on the timestep only (two GEMM rows per timestep and site, EPI_F32OUT, once per
// planned sampling loop).
// (c, rho_p) = the row's (mean, rstd) at the PREVIOUS LayerNorm site (the consumer's column tile 0 writes its own (c + mu', rho)
// for the next producer; the first site's come from primx_row_stats).  Why both:
//  * accuracy - the fold rounds (x - c) rho_p m where the reference rounds the normalised value: the same relative rounding per
//    element, but - mu' u cancels, so it costs accuracy in proportion to |mu'| / sigma (tools/ln_fold_study.py: equal to the
//    reference's rounding up to a ratio of 0.5, x 1.4 at 2, x 5 at 10).  With c = the previous site's mean, |mu'| is what ONE
//    gated branch adds to the mean.
//  * range (fp16) - (x - c) rho_p is the LayerNorm output up to the factor rho_p / rho that one gated branch changes the row's
//    spread by: O(1) whatever the magnitude or spread of the residual stream (1e4 or 1e-5 alike), as in the reference, which
//    normalises before it rounds.  Without rho_p (ABI 22) the operand carried the row's spread into the 16-bit type.
// Statistics: var = E[(x - c)^2] - mu'^2 in fp32 - harmless for the same reason.  The consumers read (c, rho_p) from one array and
// write the next pair into ANOTHER one (fold_c_out): all column tiles of a row read rho_p while tile 0 produces its successor.
// Row statistics of a consumer tile, in two steps so that the kernel can put its own memory requests between the loads and their
// use: thread t < rows owns row m0 + t.  Up to eight partials per row (fold_parts <= 8, host-checked), all loads independent.
struct FoldPartials {
    f32x2 v[8];
    f32x2 cen;                                           // (c, rho_p) the producer used
};
template <int DT>
__device__ __forceinline__ FoldPartials fold_stats_load(const GemmArgs<DT>& p, int M, int m0, int t) {
    const int m = min(m0 + t, M - 1);
    const f32x2* pp = reinterpret_cast<const f32x2*>(p.fold_part) + (int64_t)m * p.fold_parts;
    FoldPartials r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = pp[min(i, p.fold_parts - 1)];
    r.cen = reinterpret_cast<const f32x2*>(p.fold_c)[m];
    return r;
}
// stats[t] = (rho_p mu', rho / rho_p): fold_apply's y = st[1] (acc - st[0] u) + v is (rho / rho_p) acc - rho mu' u + v
template <int DT>
__device__ __forceinline__ void fold_stats_finish(const GemmArgs<DT>& p, const FoldPartials& r, int M, int K, int m0, int t,
                                                  bool tile0, f32x2* stats) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                        // fixed order
        s1 += (i < p.fold_parts) ? r.v[i][0] : 0.f;
        s2 += (i < p.fold_parts) ? r.v[i][1] : 0.f;
    }
    const float inv = 1.0f / (float)K;
    const float mu = s1 * inv;
    const float rho = 1.0f / sqrtf(fmaxf(s2 * inv - mu * mu, 0.f) + p.fold_eps);
    stats[t] = f32x2{r.cen[1] * mu, rho / r.cen[1]};     // (LDS)
    // the next producer centres with this site's mean and scales with this site's rstd
    if (tile0 && m0 + t < M) reinterpret_cast<f32x2*>(p.fold_c_out)[m0 + t] = f32x2{r.cen[0] + mu, rho};
}

// the consumer's value of four columns: st = (rho_p mu', rho / rho_p) -> (rho / rho_p) acc - rho mu' u + v
__device__ __forceinline__ f32x4 fold_apply(const f32x4 a, const f32x2 st, const f32x4 u, const f32x4 v) {
    // (two fused multiply-adds per value, as the scalar expression st[1] * (a - st[0] * u) + v contracts; packed: v_pk_fma_f32)
    const f32x4 s0 = {st[0], st[0], st[0], st[0]}, s1 = {st[1], st[1], st[1], st[1]};
    return __builtin_elementwise_fma(s1, __builtin_elementwise_fma(-s0, u, a), v);
}

// linear_out4 behind the fold: `y` already holds what bias + accumulator are there
template <int DT, bool GELU = false>
__device__ __forceinline__ typename T16<DT>::V4 fold_out4(const GemmArgs<DT>& p, const f32x4 yin) {
    using S = typename T16<DT>::S;
    if constexpr (GELU) return gelu_out4<DT>(yin);
    typename T16<DT>::V4 o;
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = rnd16<DT>(yin[j]);
    if (p.act == PRIMX_ACT_GELU_TANH) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(y[j]);
    } else if (p.act == PRIMX_ACT_GELU_ERF) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = gelu_erf_f(y[j]);
    }
    if (p.out_scale != 1.0f) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (S)(p.out_scale * rnd16<DT>(y[j]));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (S)y[j];
    }
    return o;
}

__device__ __forceinline__ float quad_sum(float v) {   // lanes 4 q .. 4 q + 3: (v0 + v1) + (v2 + v3) in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    return v;
}

template <int DT, int EPI>
__device__ __forceinline__ void epilogue_row4(const GemmArgs<DT>& p, int m, int n, const f32x4 a,
                                              const typename T16<DT>::V4 bv) {
    using S = typename T16<DT>::S;
    using V4 = typename T16<DT>::V4;
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (float)bv[j];
    }
    if (EPI == EPI_LINEAR) {
        out_store(reinterpret_cast<V4*>(p.out + (int64_t)m * p.N + n), linear_out4<DT>(p, a, bv));
    } else if (EPI == EPI_RES) {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.res) {
            const V4 rv = *reinterpret_cast<const V4*>(p.res + (int64_t)m * p.N + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = (float)rv[j];
        }
        V4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (S)((a[j] + b[j] + r[j]) * p.out_scale);
        *reinterpret_cast<V4*>(p.out + (int64_t)m * p.N + n) = o;
    } else if (EPI == EPI_GATE_RESIDUAL) {
        const int bb = m / p.rows_per_batch;
        const V4 gv = *reinterpret_cast<const V4*>(p.gate + (int64_t)bb * p.gate_stride + n);
        f32x4* xp = reinterpret_cast<f32x4*>(p.x + (int64_t)m * p.N + n);
        f32x4 xv = *xp;
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xv[j] + rnd16<DT>((float)gv[j] * rnd16<DT>(a[j] + b[j]));
        *xp = xv;
    } else if (EPI == EPI_HEADS) {  // PRIMX_HEADS_ROWS / KROWS segments only (VT segments keep the quad form)
        const int per = p.heads * p.dh;
        const int seg_all = n / per;
        const int rep = seg_all / p.n_seg, seg = seg_all - rep * p.n_seg;
        const int w = n - seg_all * per;
        const int hh = w / p.dh, dd = w - hh * p.dh;
        S* dst = (seg == 0 ? p.dst[0] : seg == 1 ? p.dst[1] : p.dst[2]) +
                 rep * (seg == 0 ? p.rep_stride[0] : seg == 1 ? p.rep_stride[1] : p.rep_stride[2]);
        const int rs = heads_row_stride(seg == 0 ? p.kind[0] : seg == 1 ? p.kind[1] : p.kind[2], p.DP);
        const int bb = m / p.rows_per_batch, tok = m - bb * p.rows_per_batch;
        V4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = rnd16<DT>(a[j] + b[j]);
            if (seg == 0 && p.scale0 != 1.0f) y = rnd16<DT>(p.scale0 * y);
            o[j] = (S)y;
        }
        *reinterpret_cast<V4*>(dst + (((int64_t)bb * p.heads + hh) * p.n_pad + tok) * rs + dd) = o;
    }
}

// MF: MFMA edge (32 -> 32x32x16, 16 -> 16x16x32); WM x WN waves, each MI x NI MFMA tiles
template <int DT, int EPI, int MF, int WM, int WN, int MI, int NI, int GATHER>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs<DT> p) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    using Acc = typename AccT<MF>::T;
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = WM * MI * MF, BN = WN * NI * MF;
    constexpr int TA = BM * 64, TW = BN * 64;                   // halves per operand tile
    constexpr int NA = (BM * 8 + 255) / 256, NW = (BN * 8 + 255) / 256;  // 16-byte chunks per thread
    constexpr int KS = (MF == 32) ? 4 : 2;                      // MFMA k-steps per BK
    constexpr int CPS = 8 / KS;                                 // 16-byte chunks per k-step
    __shared__ __attribute__((aligned(16))) S smem[2 * (TA + TW)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & (MF - 1), lg = lane / MF;             // fragment row / k-group of this lane

    const int nt = (p.N + BN - 1) / BN, mt = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, nt * mt);
    const int m0 = (id / nt) * BM, n0 = (id % nt) * BN;

    // ---- loader geometry: chunk c = tid + 256*i -> row c>>3, 16-byte column c&7
    const int kc = (tid & 7) * 8;
    const S* ga[NA];
    const S* gw[NW];
    int offa[NA], offw[NW];
    int gz[NA], gy[NA], gx[NA];  // GATHER: voxel coordinates of the row
    const S* gbase[NA];          // GATHER: &in[p, 0, 0]
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (tid + 256 * i) >> 3;
        const int ra = min(m0 + row, p.M - 1);  // clamp: rows >= M are computed but never stored
        offa[i] = lds_off(row, tid & 7);
        if (GATHER) {
            const int V = p.S3 * p.S3 * p.S3;
            const int pp = ra / V, v = ra - pp * V;
            gz[i] = v / (p.S3 * p.S3);
            gy[i] = (v / p.S3) % p.S3;
            gx[i] = v % p.S3;
            gbase[i] = p.A + ((int64_t)pp * V << p.cin_log2);
            ga[i] = nullptr;
        } else {
            ga[i] = p.A + (int64_t)ra * p.K + kc;
        }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int row = min((tid + 256 * i) >> 3, BN - 1);
        const int rw = min(n0 + row, p.N - 1);
        gw[i] = p.W + (int64_t)rw * p.K + kc;
        offw[i] = lds_off(row, tid & 7);
    }
    // the last loader iteration may be partial (T144: 1152 W chunks = 4.5 per thread; wave-uniform predicate)
    constexpr bool A_FULL = (BM * 8) % 256 == 0, W_FULL = (BN * 8) % 256 == 0;
    const bool a_last = A_FULL || (tid + 256 * (NA - 1) < BM * 8);
    const bool w_last = W_FULL || (tid + 256 * (NW - 1) < BN * 8);

    auto load_tile = [&](int kt, V8 (&ra)[NA], V8 (&rw)[NW]) {
        if (GATHER) {
            const int kk = kt * BK + kc;
            const int tap = kk >> p.cin_log2, ci = kk & ((1 << p.cin_log2) - 1);
            const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int z = gz[i] + dz, y = gy[i] + dy, x = gx[i] + dx;
                const bool ok = tap < 27 && (unsigned)z < (unsigned)p.S3 && (unsigned)y < (unsigned)p.S3 &&
                                (unsigned)x < (unsigned)p.S3;
                const int64_t off = ok ? (((int64_t)((z * p.S3 + y) * p.S3 + x) << p.cin_log2) + ci) : 0;
                ra[i] = ldg16<V8, S>(gbase[i] + off, ok);
            }
        }
        // K tail (K % 64 != 0): only the LAST k-tile can be partial.  A uniform branch keeps the per-lane predicate (compare, address
        // select, eight value selects per chunk) out of every full tile: predicated everywhere, the final Linear of the DiT
        // (4096 x 136 x 1152) ran 34.7 instead of 21 us (profiles/r4_experiments.txt section 4).  The empty asm keeps it a branch.
        const bool full = (kt + 1) * BK <= p.K;
        if (full) {
            asm volatile("" ::: "memory");
            if (!GATHER) {
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = ldg16<V8, S>(ga[i] + kt * BK, true);
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) rw[i] = ldg16<V8, S>(gw[i] + kt * BK, true);
        } else {
            asm volatile("" ::: "memory");
            const bool k_ok = kt * BK + kc < p.K;  // zero chunk past K
            if (!GATHER) {
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = ldg16<V8, S>(ga[i] + (k_ok ? kt * BK : 0), k_ok);
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) rw[i] = ldg16<V8, S>(gw[i] + (k_ok ? kt * BK : 0), k_ok);
        }
    };
    auto store_tile = [&](int buf, V8 (&ra)[NA], V8 (&rw)[NW]) {
        S* base = smem + buf * (TA + TW);
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (i < NA - 1 || a_last) *reinterpret_cast<V8*>(base + offa[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (i < NW - 1 || w_last) *reinterpret_cast<V8*>(base + TA + offw[i]) = rw[i];
    };

    Acc acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < (MF == 32 ? 16 : 4); ++r) acc[i][j][r] = 0.f;

    const int a_row = wm * MI * MF + lr, w_row = wn * NI * MF + lr;
    auto compute = [&](int buf) {
        const S* As = smem + buf * (TA + TW);
        const S* Ws = As + TA;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int chunk = s * CPS + lg;
            V8 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const V8*>(As + lds_off(a_row + i * MF, chunk));
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const V8*>(Ws + lds_off(w_row + j * MF, chunk));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if constexpr (MF == 32) acc[i][j] = T16<DT>::mfma32(a[i], b[j], acc[i][j]);
                    else acc[i][j] = T16<DT>::mfma16(a[i], b[j], acc[i][j]);
                }
        }
    };

    // ---- main loop: prefetch distance 2, two register sets, two LDS buffers.  The body is BRANCH-FREE
    // (tile indices past the end are clamped and re-load the last tile): with conditional loads hipcc's
    // wait-count pass falls back to s_waitcnt vmcnt(0) and the prefetch distance collapses to zero.
    const int nk = (p.K + BK - 1) / BK;
    V8 ra0[NA], rw0[NW], ra1[NA], rw1[NW];
    load_tile(0, ra0, rw0);
    load_tile(min(1, nk - 1), ra1, rw1);
    store_tile(0, ra0, rw0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        load_tile(min(kt + 2, nk - 1), ra0, rw0);  // set 0 is free; set 1 carries tile kt+1 in flight
        compute(0);                                // tile kt
        store_tile(1, ra1, rw1);
        __syncthreads();
        load_tile(min(kt + 3, nk - 1), ra1, rw1);  // set 1 is free; set 0 carries tile kt+2
        compute(1);                                // tile kt+1
        store_tile(0, ra0, rw0);
        __syncthreads();
    }
    if (kt < nk) compute(0);                       // odd tile count: the last tile sits in buffer 0

    // ---- epilogue.  Dense 16-bit outputs (Linear, conv + skip, k2s2 transposed conv) with N % 8 == 0: the tile is parked
    // in LDS as fp32 [BM][BN] (exactly the operand buffers' 64 KB for the 128x128 tile) and walked ROW-MAJOR, one
    // (row, 8 consecutive columns) unit per thread per pass: 16-byte stores, whole 256-byte rows per 16 lanes - the quad
    // form below writes 2-byte scalars in 64-byte runs (the k2s2 upsample, K = 256 and 537 MB of output, ran at 0.6 TB/s).
    if constexpr (MF == 32 && (EPI == EPI_LINEAR || EPI == EPI_RES || EPI == EPI_CONVT)) {
        if ((p.N & 7) == 0 && (EPI != EPI_CONVT || (p.cout & 7) == 0)) {
            static_assert(BM * BN * 2 <= 2 * (TA + TW), "fp32 tile must fit the operand buffers");
            float* red = reinterpret_cast<float*>(smem);
            __syncthreads();                                   // every wave is done with the operand buffers
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * BN + (wn * NI + ni) * 32 + lr] = acc[mi][ni][r];
            __syncthreads();
            using V8o = typename T16<DT>::V8;
            using V4o = typename T16<DT>::V4;
            constexpr int UPR = BN / 8;                        // units per row
            for (int u = tid; u < BM * UPR; u += 256) {
                const int row = u / UPR, c8 = u - row * UPR;
                const int m = m0 + row, n = n0 + 8 * c8;
                if (m >= p.M || n >= p.N) continue;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(red + row * BN + 8 * c8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(red + row * BN + 8 * c8 + 4);
                V8o o;
                if (EPI == EPI_LINEAR) {
                    V4o b0 = V4o{}, b1 = V4o{};
                    if (p.bias) { b0 = *reinterpret_cast<const V4o*>(p.bias + n); b1 = *reinterpret_cast<const V4o*>(p.bias + n + 4); }
                    const V4o o0 = linear_out4<DT>(p, v0, b0), o1 = linear_out4<DT>(p, v1, b1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = o0[e]; o[4 + e] = o1[e]; }
                    *reinterpret_cast<V8o*>(p.out + (int64_t)m * p.N + n) = o;
                } else if (EPI == EPI_RES) {
                    V8o bv = V8o{}, rv = V8o{};
                    if (p.bias) bv = *reinterpret_cast<const V8o*>(p.bias + n);
                    if (p.res) rv = *reinterpret_cast<const V8o*>(p.res + (int64_t)m * p.N + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float y = (e < 4 ? v0[e] : v1[e - 4]) + (p.bias ? (float)bv[e] : 0.f);
                        if (p.res) y += (float)rv[e];
                        o[e] = (S)(y * p.out_scale);
                    }
                    *reinterpret_cast<V8o*>(p.out + (int64_t)m * p.N + n) = o;
                } else {  // EPI_CONVT: row m = (prim, z, y, x), columns = (tap ; 8 consecutive co) -> voxel (2z+dz, 2y+dy, 2x+dx)
                    const int Sg = p.S3, V = Sg * Sg * Sg, S2 = 2 * Sg;
                    const int seg = n / p.cout, dd = n - seg * p.cout;
                    const int dz = seg >> 2, dy = (seg >> 1) & 1, dx = seg & 1;
                    const int pp = m / V, v = m - pp * V;
                    const int z = v / (Sg * Sg), y = (v / Sg) % Sg, x = v % Sg;
                    const int64_t ov = ((int64_t)(2 * z + dz) * S2 + (2 * y + dy)) * S2 + (2 * x + dx);
                    V8o bv = V8o{};
                    if (p.bias) bv = *reinterpret_cast<const V8o*>(p.bias + dd);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (S)((e < 4 ? v0[e] : v1[e - 4]) + (p.bias ? (float)bv[e] : 0.f));
                    *reinterpret_cast<V8o*>(p.out + ((int64_t)pp * 8 * V + ov) * p.cout + dd) = o;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const ColInfo c = make_col<DT, EPI>(p, n0 + (wn * NI + ni) * MF + lr);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if constexpr (MF == 32) {
                // acc[r] = C[tile_m + (r&3) + 8*(r>>2) + 4*lg][tile_n + lr]
                const int mbase = m0 + (wm * MI + mi) * 32 + 4 * lg;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float q[4] = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2],
                                        acc[mi][ni][4 * g + 3]};
                    epilogue_quad<DT, EPI>(p, c, mbase + 8 * g, q);
                }
            } else {
                // acc[r] = C[tile_m + 4*lg + r][tile_n + lr]
                const float q[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                epilogue_quad<DT, EPI>(p, c, m0 + (wm * MI + mi) * 16 + 4 * lg, q);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// T144: 128 x 144 tile, 8 waves = 4 (M) x 2 (K halves).  Wave (kg, wm) multiplies rows wm*32..+31 by all 144 columns over k-half kg
// (32 of the 64 k's of every k-tile) with 2 x 9 tiles of 16x16x32; the two halves meet in LDS.  (A register-staged form of this tile
// served K tails until round 4; those launches - small test models only - now take the generic 128 x 128 kernel above.)
// T144 with LDS-DMA staging (K % 64 == 0): same tile / wave roles / reduction as described above, but the
// operand tiles go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a
// 3-stage ring: no staging VGPRs, no ds_write pass (272 of the 640 LDS-array cycles per k-tile measured on
// the register-staged kernel, profiles/r1_gemm_pmc.txt), two tiles in flight across every barrier.
// A stage is one 272-row x 128-byte image (rows 0..127 = A, 128..271 = W); wave-instruction t covers rows
// 8t..8t+7, and because LDS-DMA writes lane-linearly (base + lane*16) the XOR swizzle is applied to the
// per-lane SOURCE address: lane (r = lane>>3, c' = lane&7) fetches global chunk c' ^ ((row>>1)&7) of its row
// (cdna_hip_programming.md rule 21).  Sync per k-tile: s_waitcnt vmcnt(4) (this wave's DMAs of tile kt have
// landed, tile kt+1's stay in flight) + raw s_barrier; __syncthreads() would drain the ring.
// timeline profile (PRIMX_GEMM_PROF=1): [0] min start, [1] max end (s_memrealtime, 100 MHz), sums of core-clock cycles:
// [2] entry -> tile 0 landed, [3] main loop, [4] epilogue, [5] workgroups, [6] sum of (start - min start) in 10 ns ticks,
// [8] / [9] sums of each workgroup's lifetime in 10 ns ticks / in core cycles (their ratio = the shader clock under this kernel)
__device__ unsigned long long g_gemm_prof[12];
// per-workgroup records of the same launch (first 4096 workgroups): {start tick - launch minimum is taken on the host, lifetime
// in ticks, lifetime in core cycles, (XCC id << 32) | epilogue cycles} - the kernel ends with its SLOWEST workgroup
__device__ unsigned long long g_gemm_wg[4096][4];

template <int DT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm144_dma_kernel(PRIMX_GEMM_PARAMS(DT)) {
    PRIMX_GEMM_ARGS(DT);
    unsigned long long pr0 = 0, pc0 = 0, pc1 = 0, pc2 = 0;
    if (pl_prof) { pr0 = __builtin_amdgcn_s_memrealtime(); pc0 = __builtin_readcyclecounter(); }
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    typedef __attribute__((address_space(1))) const void GV;
    typedef __attribute__((address_space(3))) void LV;
    constexpr int BM = 128, BN = 144, MI = 2, NI = 9;
    constexpr int ROWS = BM + BN;                        // 272 rows per stage
    constexpr int STAGE = ROWS * 64;                     // halves per stage (34,816 B)
    constexpr int NINST = ROWS / 8;                      // 34 wave-instructions per stage
    constexpr int NSLOT = (NINST + 7) / 8;               // 5 slots per wave (waves 0,1 use all 5, others 4)
    constexpr int RED_HALVES = BM * BN * 2;
    constexpr int RS = BN + 4;                           // fp32 row stride of the row-major epilogue staging
    constexpr int ROWMAJOR_HALVES = 2 * BM * RS * 2;     // both K halves, fp32 (151,552 B)
    constexpr int NST = 3;                               // ring depth: NST - 1 tiles in flight across every barrier (4 stages measured the same)
    constexpr int LDS_HALVES = (NST * STAGE > ROWMAJOR_HALVES) ? NST * STAGE : ROWMAJOR_HALVES;
    static_assert(LDS_HALVES * 2 <= 160 * 1024 && RED_HALVES <= LDS_HALVES, "LDS budget");
    __shared__ __attribute__((aligned(16))) S smem[LDS_HALVES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;

    const int nt = pl_N / BN, mt = (pl_M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, nt * mt);
    const int m0 = (id / nt) * BM, n0 = (id % nt) * BN;

    // per-slot source pointers (k-tile 0) for this lane
    const S* gp[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int t = min(wave + 8 * i, NINST - 1);
        const int row = 8 * t + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        gp[i] = (row < BM) ? pl_A + (int64_t)min(m0 + row, pl_M - 1) * pl_K + c * 8
                           : pl_W + (int64_t)(n0 + row - BM) * pl_K + c * 8;
    }
    const bool last_slot = wave + 8 * (NSLOT - 1) < NINST;  // wave-uniform
    auto issue_one = [&](int kt, int stage, int i) {        // instruction t = wave + 8 i lands at stage + t * 1 KiB
        if (i < NSLOT - 1 || last_slot)
            __builtin_amdgcn_global_load_lds((GV*)(uintptr_t)(gp[i] + kt * BK),
                                             (LV*)(smem + stage * STAGE + wave * 512 + i * 8 * 512), 16, 0, 0);
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) issue_one(kt, stage, i);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Epilogue form (tile-uniform).  quad_form: column tiles of a PRIMX_HEADS_VT segment (4 consecutive TOKENS per lane is
    // what that layout wants) or tiles straddling two head segments; otherwise the LDS row-major walk below.  (A register
    // epilogue with swapped MFMA operands - what the 256x288 kernel uses - measured no better here and was removed in round 3.)
    bool quad_form = false;
    if (EPI == EPI_HEADS) {
        const int per = p.heads * p.dh;
        quad_form = (per % BN != 0) || p.kind[(n0 / per) % p.n_seg] == PRIMX_HEADS_VT;
    }

    const int a_row = wm * 32 + lr;
    const int chunk = kg * 4 + lg;
    // Fragment double buffering: tile kt+1's fragments are read from LDS (non-blocking ds_read_b128) BEFORE the
    // 18 MFMAs of tile kt issue, so the LDS-read phase of one tile overlaps the MFMA phase of the previous one
    // inside every wave (the lock-step "read all, then multiply" form left the MFMA pipe 58 % idle).
    auto read_frags = [&](int stage, V8 (&a)[MI], V8 (&b)[NI]) {
        const S* As = smem + stage * STAGE;
        const S* Ws = As + BM * 64;
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const V8*>(As + lds_off(a_row + i * 16, chunk));
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const V8*>(Ws + lds_off(lr + j * 16, chunk));
    };
    auto multiply = [&](const V8 (&a)[MI], const V8 (&b)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = T16<DT>::mfma16(a[i], b[j], acc[i][j]);
    };

    // Ring: tile j lives in stage j % 3.  Step kt: [tile kt+1 landed, everyone done reading tile kt] ->
    // DMA tile kt+3 into tile kt's stage -> read fragments of tile kt+1 -> MFMAs of tile kt.
    // EPI_GATE_RESIDUAL: the fp32 residual rows this thread will update in the row-major epilogue are fetched NOW
    // (9 x 16 bytes per thread); they land during the main loop, so the epilogue only has to add and write.
    constexpr int NROWCH = (BM * (BN / 4)) / 512;
    f32x4 xpre[NROWCH];
    if (EPI == EPI_GATE_RESIDUAL) {
#pragma unroll
        for (int i = 0; i < NROWCH; ++i) {
            const int cid = tid + 512 * i;
            const int row = cid / (BN / 4), c4 = cid - row * (BN / 4);
            const int m = min(m0 + row, pl_M - 1);
            xpre[i] = *reinterpret_cast<const f32x4*>(p.x + (int64_t)m * pl_N + n0 + 4 * c4);
        }
    }

    const int nk = pl_K / BK;
#pragma unroll
    for (int s0 = 0; s0 < NST; ++s0) issue(min(s0, nk - 1), s0);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (NST - 1)) : "memory");  // tile 0 landed (4..5 DMAs per tile per wave)
    if (pl_prof) pc1 = __builtin_readcyclecounter();
    V8 a0[MI], b0[NI], a1[MI], b1[NI];
    read_frags(0, a0, b0);
    int st_cur = 0, st_next = 1;  // stage of tile kt / tile kt+1
    auto step = [&](int kt, const V8 (&ac)[MI], const V8 (&bc)[NI], V8 (&an)[MI], V8 (&bn)[NI]) {
        // vmcnt(4): tile kt+1 landed for this wave (tile kt+2 may stay in flight); lgkmcnt(0): this wave's reads of
        // tile kt's stage have completed, so after the barrier that stage can be overwritten by the DMA below
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(4 * (NST - 2)) : "memory");
        // (Spreading these DMA issues between the MFMA groups, which pays in the 256x288 kernel, measured worse here:
        // main loop 78.2k -> 81.8k cycles at K = 4608.)
        issue(min(kt + NST, nk - 1), st_cur);
        read_frags(st_next, an, bn);
        multiply(ac, bc);
        st_cur = st_next;
        st_next = (st_next == NST - 1) ? 0 : st_next + 1;
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        step(kt, a0, b0, a1, b1);
        step(kt + 1, a1, b1, a0, b0);
    }
    if (kt < nk) step(kt, a0, b0, a1, b1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // drain the (redundant) tail DMAs before LDS reuse
    if (pl_prof) pc2 = __builtin_readcyclecounter();
    unsigned long long pc_stg = 0;
    auto prof_end = [&]() {
        if (pl_prof) {
            const unsigned long long pc_iss = __builtin_readcyclecounter();
            if (tid == 0) atomicAdd(&g_gemm_prof[7], ((pc_stg - pc2) << 32) | (pc_iss - pc_stg));
            __builtin_amdgcn_s_waitcnt(0);   // the epilogue's stores have been issued AND acknowledged
            const unsigned long long pc3 = __builtin_readcyclecounter(), pr1 = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) {
                atomicMin(&g_gemm_prof[0], pr0); atomicMax(&g_gemm_prof[1], pr1);
                atomicAdd(&g_gemm_prof[2], pc1 - pc0); atomicAdd(&g_gemm_prof[3], pc2 - pc1);
                atomicAdd(&g_gemm_prof[4], pc3 - pc2); atomicAdd(&g_gemm_prof[5], 1ull); atomicAdd(&g_gemm_prof[6], pr0);
                atomicAdd(&g_gemm_prof[8], pr1 - pr0); atomicAdd(&g_gemm_prof[9], pc3 - pc0);   // -> the shader clock while this kernel runs
                if (blockIdx.x < 4096) {
                    unsigned xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    g_gemm_wg[blockIdx.x][0] = pr0; g_gemm_wg[blockIdx.x][1] = pr1 - pr0; g_gemm_wg[blockIdx.x][2] = pc3 - pc0;
                    g_gemm_wg[blockIdx.x][3] = ((unsigned long long)(xcc & 15) << 32) | (unsigned)(pc3 - pc2);
                }
            }
        }
    };

    // ---- epilogue.  Default: both K halves park their accumulators in LDS as fp32 [half][128][148]; after one
    // barrier all 512 threads walk the tile ROW-MAJOR, 4 consecutive columns each (sum of the two halves ->
    // epilogue_row4): coalesced 8 / 16-byte accesses instead of 2 / 4-byte ones (the fp32 residual read-modify-write
    // alone cost +12 us per launch in quad form, tools/gemm_ksweep.py).  Column tiles that belong to a
    // PRIMX_HEADS_VT segment keep the quad form (4 consecutive TOKENS per lane is what that layout wants).
    float* red = reinterpret_cast<float*>(smem);
    // EPI_HEADS fast path: the tile lies inside ONE (repetition, segment), so everything that needs a division is
    // tile-uniform and computed once on the scalar unit; per unit only compares remain (the generic
    // epilogue_row4 does five integer divisions per unit: +2.5 us per round, PRIMX_GEMM_PROF).
    int h_hh0 = 0, h_dd0 = 0, h_bb0 = 0, h_tok0 = 0, h_rs = 0, h_seg = 0;
    S* h_dst = nullptr;
    bool h_fast = false;
    if (EPI == EPI_HEADS) {
        h_fast = p.dh >= 48 && (p.dh & 3) == 0 && p.rows_per_batch >= BM;
        const int per = p.heads * p.dh;
        const int seg_all = n0 / per, rep_i = seg_all / p.n_seg;
        h_seg = seg_all - rep_i * p.n_seg;
        const int w0 = n0 - seg_all * per;
        h_hh0 = w0 / p.dh;
        h_dd0 = w0 - h_hh0 * p.dh;
        h_bb0 = m0 / p.rows_per_batch;
        h_tok0 = m0 - h_bb0 * p.rows_per_batch;
        h_rs = heads_row_stride(h_seg == 0 ? p.kind[0] : h_seg == 1 ? p.kind[1] : p.kind[2], p.DP);
        h_dst = (h_seg == 0 ? p.dst[0] : h_seg == 1 ? p.dst[1] : p.dst[2]) +
                rep_i * (h_seg == 0 ? p.rep_stride[0] : h_seg == 1 ? p.rep_stride[1] : p.rep_stride[2]);
    }
    if (!quad_form) {
        // bias / gate vectors of this thread's 9 row-chunks: loaded now, they land under the LDS staging below
        using V4e = typename T16<DT>::V4;
        V4e bpre[NROWCH], gpre[NROWCH];
#pragma unroll
        for (int i = 0; i < NROWCH; ++i) {
            const int cid = tid + 512 * i;
            const int row = cid / (BN / 4), c4 = cid - row * (BN / 4);
            bpre[i] = V4e{};
            if (p.bias && EPI != EPI_CONVT) bpre[i] = *reinterpret_cast<const V4e*>(p.bias + n0 + 4 * c4);
            if (EPI == EPI_GATE_RESIDUAL) {
                const int m = min(m0 + row, pl_M - 1);
                gpre[i] = *reinterpret_cast<const V4e*>(p.gate + (int64_t)(m / p.rows_per_batch) * p.gate_stride + n0 + 4 * c4);
            }
        }
        float* mine = red + kg * (BM * RS);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(wm * 32 + mi * 16 + 4 * lg + r) * RS + ni * 16 + lr] = acc[mi][ni][r];
        __syncthreads();
        if (pl_prof) pc_stg = __builtin_readcyclecounter();
        // (Issuing all 18 LDS reads up front - hipcc sinks each pair into the guarded block that uses it - shortens the
        // issue phase 6.2k -> 4.0k cycles but not the kernel: the epilogue ends when the stores are acknowledged.)
#pragma unroll
        for (int i = 0; i < (BM * (BN / 4)) / 512; ++i) {   // 4608 row-chunks / 512 threads = 9
            const int cid = tid + 512 * i;
            const int row = cid / (BN / 4), c4 = cid - row * (BN / 4);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(red + row * RS + 4 * c4);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(red + BM * RS + row * RS + 4 * c4);
            if (m0 + row >= pl_M) continue;
            if (EPI == EPI_HEADS && h_fast) {
                using V4 = typename T16<DT>::V4;
                int d = h_dd0 + 4 * c4, hh = h_hh0;          // d < dh + 144 <= 4 dh
                if (d >= p.dh) { d -= p.dh; ++hh; }
                if (d >= p.dh) { d -= p.dh; ++hh; }
                if (d >= p.dh) { d -= p.dh; ++hh; }
                int tok = h_tok0 + row, bb = h_bb0;          // tok < rows_per_batch + 128 <= 2 rows_per_batch
                if (tok >= p.rows_per_batch) { tok -= p.rows_per_batch; ++bb; }
                const V4 bv = bpre[i];
                V4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y = rnd16<DT>(v0[j] + v1[j] + (float)bv[j]);
                    if (h_seg == 0 && p.scale0 != 1.0f) y = rnd16<DT>(p.scale0 * y);
                    o[j] = (S)y;
                }
                out_store(reinterpret_cast<V4*>(h_dst + (((int64_t)bb * p.heads + hh) * p.n_pad + tok) * h_rs + d), o);
            } else if (EPI == EPI_GATE_RESIDUAL) {
                using V4 = typename T16<DT>::V4;
                const int m = m0 + row, n = n0 + 4 * c4;
                const V4 gv = gpre[i], bv = bpre[i];
                f32x4 xv = xpre[i];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xv[j] = xv[j] + rnd16<DT>((float)gv[j] * rnd16<DT>(v0[j] + v1[j] + (float)bv[j]));
                out_store(reinterpret_cast<f32x4*>(p.x + (int64_t)m * pl_N + n), xv);
            } else {
                epilogue_row4<DT, EPI>(p, m0 + row, n0 + 4 * c4, v0 + v1, bpre[i]);
            }
        }
        prof_end();
        return;
    }
    // ---- quad form: K-half 0 owns column tiles 0..4, K-half 1 owns 5..8
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const bool mine = (ni < 5) == (kg == 0);
            if (!mine) {
                float* dst = red + (((wm * MI + mi) * NI + ni) * 4) * 64 + lane;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r * 64] = acc[mi][ni][r];
            }
        }
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const bool mine = (ni < 5) == (kg == 0);
        if (!mine) continue;
        const ColInfo c = make_col<DT, EPI>(p, n0 + ni * 16 + lr);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const float* src = red + (((wm * MI + mi) * NI + ni) * 4) * 64 + lane;
            const float q[4] = {acc[mi][ni][0] + src[0], acc[mi][ni][1] + src[64], acc[mi][ni][2] + src[128],
                                acc[mi][ni][3] + src[192]};
            epilogue_quad<DT, EPI>(p, c, m0 + wm * 32 + mi * 16 + 4 * lg, q);
        }
    }
    prof_end();
}

// spin waits of gemm144l_dma_kernel<., EPI_GATE_RESIDUAL_LN> that gave up (primx_ln_sync_timeouts): must stay 0
__device__ unsigned g_ln_sync_timeouts = 0;

// ---------------------------------------------------------------------------------------------------
// The 128 x 144 LDS-DMA kernel with LOADER WAVES (PRIMX_GEMM_LOADER=0 switches it off): the dense-output epilogues and the
// token-major heads epilogue (host-checked: tiles inside one segment, no PRIMX_HEADS_VT segment, dh >= 48, dh % 4 == 0,
// rows_per_batch >= 128 - launch144_dma).
// Why: in gemm144_dma_kernel every wave issues its 4-5 DMA instructions behind the step barrier and then sits in the
// issue queue until the unit has taken them (24.5 cycles per 1 KiB instruction, 833 per k-tile for the workgroup,
// PRIMX_GEMM_PROF=2) - its 18 MFMAs start late, and a step costs 1096 cycles instead of max(833, 576).  Here waves 8 and 9
// do nothing but issue DMA (17 instructions per tile each) and wait for it; the eight compute waves never touch the
// vector-memory queue inside the loop.  10 waves = 3 on two of the SIMDs: 168 VGPRs per wave.
// Barrier protocol (all 10 waves execute every s_barrier): P (tile 0 landed) | S_kt per step (tile kt+1 landed: the loaders
// waited for their own pieces; all fragment reads of tile kt are home: the compute waves waited lgkmcnt(0)) | D (ring
// drained) | E (accumulators parked for the row-major walk).
template <int DT, int EPI>
__global__ __launch_bounds__(640) void gemm144l_dma_kernel(PRIMX_GEMM_PARAMS(DT)) {
    PRIMX_GEMM_ARGS(DT);
    unsigned long long pr0 = 0, pc0 = 0, pc1 = 0, pc2 = 0, pc_stg = 0, pc_iss = 0;   // PRIMX_GEMM_PROF=1 timeline of compute wave 0 (see g_gemm_prof)
    if (pl_prof) { pr0 = __builtin_amdgcn_s_memrealtime(); pc0 = __builtin_readcyclecounter(); }
    static_assert(EPI == EPI_LINEAR || EPI == EPI_GATE_RESIDUAL || EPI == EPI_HEADS || EPI == EPI_GATE_RESIDUAL_LN ||
                  EPI == EPI_GATE_RESIDUAL_FOLD || EPI == EPI_HEADS_FOLD || EPI == EPI_LINEAR_FOLD || EPI == EPI_F32OUT,
                  "row-major epilogues only");
    constexpr bool GATE_RES = EPI == EPI_GATE_RESIDUAL || EPI == EPI_GATE_RESIDUAL_LN || EPI == EPI_GATE_RESIDUAL_FOLD;
    constexpr bool HEADS = EPI == EPI_HEADS || EPI == EPI_HEADS_FOLD;
    constexpr bool FOLD_P = EPI == EPI_GATE_RESIDUAL_FOLD;                          // producer of a folded LayerNorm site
    constexpr bool FOLD_C = EPI == EPI_HEADS_FOLD || EPI == EPI_LINEAR_FOLD;        // consumer
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    using V4e = typename T16<DT>::V4;
    typedef __attribute__((address_space(1))) const void GV;
    typedef __attribute__((address_space(3))) void LV;
    constexpr int BM = 128, BN = 144, MI = 2, NI = 9, NST = 3;   // (4 stages fit and measured no better, also with cold weights in the step: 9.31 -> 9.36 ms)
    constexpr int ROWS = BM + BN, STAGE = ROWS * 64, NINST = ROWS / 8, NL = NINST / 2;   // 17 wave-instructions per loader per tile
    static_assert((NST - 1) * NL <= 63, "vmcnt is a 6-bit counter");
    constexpr int RS = BN + 4;
    constexpr int ROWMAJOR_HALVES = 2 * BM * RS * 2;
    constexpr int LDS_HALVES = (NST * STAGE > ROWMAJOR_HALVES) ? NST * STAGE : ROWMAJOR_HALVES;
    // fold consumer, behind the ring: (mu', rho) of the tile's 128 rows, then u and v of its 144 columns - fetched ONCE per workgroup
    // at kernel start (every thread loading the vectors of its nine units from L2 was 147 KB through the L1 per workgroup: to_q
    // 19.2 vs 17.7 us)
    constexpr int STAT_HALVES = FOLD_C ? BM * 4 + 2 * BN * 2 : 0;
    static_assert((LDS_HALVES + STAT_HALVES) * 2 <= 160 * 1024 && NINST % 2 == 0, "LDS budget / loader split");
    __shared__ __attribute__((aligned(16))) S smem[LDS_HALVES + STAT_HALVES];
    // (Round 2 left an untested option here - the LOADER waves fetching 5 of the 9 fp32 residual row-chunks of every thread by
    // LDS-DMA into the 47 KB behind the ring.  Measured in round 3, same box: gate-residual 19.3 / 19.5 vs 19.2 / 18.4 us at
    // K = 1152, 49.6 / 47.8 vs 46.6 / 49.3 us at K = 4608, the configs[1] step 9.23 - 9.27 vs 9.12 - 9.17 ms: slower.  vmcnt is ONE
    // in-order counter per wave: the residual rows come from MALL / HBM, the tiles from L2, and every counted "tile landed" wait
    // of the loader also waited for the slower loads queued before it.  The same rows requested by the COMPUTE waves before the
    // main loop (their vmcnt queue is never waited on in the loop; 4 / 6 / 8 of the 9 chunks in 16 / 24 / 32 more VGPRs, 32-bit
    // offsets, no scratch) measured no better either: gate-residual 19.9 - 20.8 vs 19.8 - 20.3 us at K = 1152, the step 9.23 - 9.26 /
    // 9.30 - 9.34 / 9.30 - 9.33 vs 9.19 - 9.24 ms (tools/gpu/r3_s2.sh).  The epilogue does not wait for the residual READ: the
    // timeline of the Linear epilogue, which reads nothing, already shows it - 2.6k cycles of parking, 4.1k of store issue and
    // 4.7k until the last store is acknowledged, i.e. the 256 workgroups' simultaneous WRITE burst.  Both options removed.)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = pl_N / BN, mt = (pl_M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, nt * mt);
    const int m0 = (id / nt) * BM, n0 = (id % nt) * BN;
    const int nk = pl_K / BK;

    if (wave >= 8) {
        if (PRIMX_LOADER_PRIO) __builtin_amdgcn_s_setprio(PRIMX_LOADER_PRIO);
        // ---------------- loader wave lw: instructions t = lw * 17 + i, rows 8t .. 8t+7 of the 272-row stage image
        const int lw = wave - 8;
        const S* gp[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int row = 8 * (lw * NL + i) + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            gp[i] = (row < BM) ? pl_A + (int64_t)min(m0 + row, pl_M - 1) * pl_K + c * 8
                               : pl_W + (int64_t)(n0 + row - BM) * pl_K + c * 8;
        }
        auto issue = [&](int kt, int stage) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                __builtin_amdgcn_global_load_lds((GV*)(uintptr_t)(gp[i] + kt * BK),
                                                 (LV*)(smem + stage * STAGE + (lw * NL + i) * 512), 16, 0, 0);
        };
#pragma unroll
        for (int s0 = 0; s0 < NST; ++s0) issue(min(s0, nk - 1), s0);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST - 1) * NL) : "memory");   // P: tile 0 landed
        int st = 0;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST - 2) * NL) : "memory");   // S_kt: tile kt+1 landed, NST - 2 newer ones may fly
            issue(min(kt + NST, nk - 1), st);                                              // tile kt's stage is free now
            st = (st == NST - 1) ? 0 : st + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                   // D
        asm volatile("s_barrier" ::: "memory");                                          // E
        if constexpr (FOLD_P) asm volatile("s_barrier" ::: "memory");                    // H (every wave executes every barrier)
        if constexpr (EPI == EPI_GATE_RESIDUAL_LN) {
            if (pl_rest.ln_light & 8) asm volatile("s_barrier\n\ts_barrier" ::: "memory");   // F, G of the LayerNorm tail (every wave executes every barrier)
        }
        return;
    }

    // ---------------- compute waves: the tile / wave roles of gemm144_dma_kernel
    if (PRIMX_COMPUTE_PRIO) __builtin_amdgcn_s_setprio(PRIMX_COMPUTE_PRIO);
    const int kg = wave >> 2, wm = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_row = wm * 32 + lr;
    const int chunk = kg * 4 + lg;
    auto read_frags = [&](int stage, V8 (&a)[MI], V8 (&b)[NI]) {
        const S* As = smem + stage * STAGE;
        const S* Ws = As + BM * 64;
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const V8*>(As + lds_off(a_row + i * 16, chunk));
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const V8*>(Ws + lds_off(lr + j * 16, chunk));
    };
    auto multiply = [&](const V8 (&a)[MI], const V8 (&b)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = T16<DT>::mfma16(a[i], b[j], acc[i][j]);
    };
    constexpr int NROWCH = (BM * (BN / 4)) / 512;
    f32x4 xpre[NROWCH];
    // fold consumer: the tile's row statistics from the producer's partial sums, while the loaders fetch the first tiles (requested
    // BEFORE the prefetch lines: vmcnt is in-order, the wait for these must not include those)
    f32x2* const fstat = reinterpret_cast<f32x2*>(smem + LDS_HALVES);
    float* const fu = reinterpret_cast<float*>(smem + LDS_HALVES + BM * 4);              // u[144], then v[144]
    FoldPartials fpart;
    f32x2 fu2 = {0.f, 0.f}, fv2 = {0.f, 0.f};
    const int tuv = tid - BM;                                                            // threads 128 .. 199: two columns each
    if constexpr (FOLD_C) {
        if (wave < BM / 64) fpart = fold_stats_load<DT>(p, pl_M, m0, tid);
        else if (tuv < BN / 2) {
            fu2 = *reinterpret_cast<const f32x2*>(p.fold_u + n0 + 2 * tuv);
            fv2 = *reinterpret_cast<const f32x2*>(p.fold_v + n0 + 2 * tuv);
        }
    }
    const pf_u32x2 pf_v = gemm_prefetch_lines<DT>(p, wave, lane);                        // (the launch's prefetch range)
    if constexpr (FOLD_C) {
        if (wave < BM / 64) fold_stats_finish<DT>(p, fpart, pl_M, pl_K, m0, tid, n0 == 0, fstat);
        else if (tuv < BN / 2) {
            *reinterpret_cast<f32x2*>(fu + 2 * tuv) = fu2;
            *reinterpret_cast<f32x2*>(fu + BN + 2 * tuv) = fv2;
        }
    }
    asm volatile("s_barrier" ::: "memory");                                              // P
    if (pl_prof) pc1 = __builtin_readcyclecounter();
    V8 a0[MI], b0[NI], a1[MI], b1[NI];
    read_frags(0, a0, b0);
    int st_next = 1;
    auto step = [&](const V8 (&ac)[MI], const V8 (&bc)[NI], V8 (&an)[MI], V8 (&bn)[NI]) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                 // S_kt
        read_frags(st_next, an, bn);
        multiply(ac, bc);
        st_next = (st_next == NST - 1) ? 0 : st_next + 1;
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        step(a0, b0, a1, b1);
        step(a1, b1, a0, b0);
    }
    if (kt < nk) step(a0, b0, a1, b1);
    asm volatile("" ::"v"(pf_v[0]), "v"(pf_v[1]));                                                        // the prefetch requests have returned
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                     // D: the stages may be reused
    if (pl_prof) pc2 = __builtin_readcyclecounter();

    // ---------------- epilogue: both K halves park their accumulators as fp32 [half][128][148]; row-major walk, 4 columns
    // per thread, 9 row-chunks each (the form of gemm144_dma_kernel).  The residual / gate / bias vectors are
    // requested AFTER the parking (the accumulator registers are free by then: held across it they spilled) and land under
    // the barrier
    float* red = reinterpret_cast<float*>(smem);
    float* mine = red + kg * (BM * RS);
    // (72 ds_write_b32 per lane.  With the MFMA operands swapped the accumulator holds C^T and a tile parks with 18 ds_write_b128 -
    // measured in round 3, same box: gate-residual 18.8 / 48.4 vs 19.1 / 47.5 us, the step 9.06 - 9.13 vs 8.90 - 8.97 ms: slower
    // (16 lanes write the same 4 columns of 16 rows, 592 bytes apart: the row stride that suits the walk below conflicts there).)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(wm * 32 + mi * 16 + 4 * lg + r) * RS + ni * 16 + lr] = acc[mi][ni][r];
    __builtin_amdgcn_sched_barrier(0);
    // EPI_HEADS: the tile lies inside ONE (repetition, segment): everything that needs a division is tile-uniform
    int h_hh0 = 0, h_dd0 = 0, h_bb0 = 0, h_tok0 = 0, h_rs = 0, h_seg = 0;
    S* h_dst = nullptr;
    if (HEADS) {
        const int per = p.heads * p.dh;
        const int seg_all = n0 / per, rep_i = seg_all / p.n_seg;
        h_seg = seg_all - rep_i * p.n_seg;
        const int w0 = n0 - seg_all * per;
        h_hh0 = w0 / p.dh;
        h_dd0 = w0 - h_hh0 * p.dh;
        h_bb0 = m0 / p.rows_per_batch;
        h_tok0 = m0 - h_bb0 * p.rows_per_batch;
        h_rs = heads_row_stride(h_seg == 0 ? p.kind[0] : h_seg == 1 ? p.kind[1] : p.kind[2], p.DP);
        h_dst = (h_seg == 0 ? p.dst[0] : h_seg == 1 ? p.dst[1] : p.dst[2]) +
                rep_i * (h_seg == 0 ? p.rep_stride[0] : h_seg == 1 ? p.rep_stride[1] : p.rep_stride[2]);
    }
    V4e bpre[NROWCH], gpre[NROWCH];
    V4e spre[FOLD_P ? NROWCH : 1];             // fold producer: the next LayerNorm's scale vector, the row's (centre, scale)
    f32x2 cpre[FOLD_P ? NROWCH : 1];
#pragma unroll
    for (int i = 0; i < NROWCH; ++i) {
        const int cid = tid + 512 * i;
        const int row = cid / (BN / 4), c4 = cid - row * (BN / 4);
        const int m = min(m0 + row, pl_M - 1);
        bpre[i] = V4e{};
        if (!FOLD_C && p.bias) bpre[i] = *reinterpret_cast<const V4e*>(p.bias + n0 + 4 * c4);
        if (GATE_RES) {
            gpre[i] = *reinterpret_cast<const V4e*>(p.gate + (int64_t)(m / p.rows_per_batch) * p.gate_stride + n0 + 4 * c4);
            xpre[i] = *reinterpret_cast<const f32x4*>(p.x + (int64_t)m * pl_N + n0 + 4 * c4);
            if constexpr (FOLD_P) {
                spre[i] = *reinterpret_cast<const V4e*>(p.ln_scale + (int64_t)(m / p.rows_per_batch) * p.ln_mod_stride + n0 + 4 * c4);
                cpre[i] = reinterpret_cast<const f32x2*>(p.fold_c)[m];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                     // E
    if (pl_prof) pc_stg = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < NROWCH; ++i) {
        const int cid = tid + 512 * i;
        const int row = cid / (BN / 4), c4 = cid - row * (BN / 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(red + row * RS + 4 * c4);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(red + BM * RS + row * RS + 4 * c4);
        if (m0 + row >= pl_M) continue;
        if (HEADS) {
            int d = h_dd0 + 4 * c4, hh = h_hh0;          // d < dh + 144 <= 4 dh
            if (d >= p.dh) { d -= p.dh; ++hh; }
            if (d >= p.dh) { d -= p.dh; ++hh; }
            if (d >= p.dh) { d -= p.dh; ++hh; }
            int tok = h_tok0 + row, bb = h_bb0;          // tok < rows_per_batch + 128 <= 2 rows_per_batch
            if (tok >= p.rows_per_batch) { tok -= p.rows_per_batch; ++bb; }
            const V4e bv = bpre[i];
            V4e o;
            if constexpr (FOLD_C) {
                const f32x4 pre = fold_apply(v0 + v1, fstat[row], *reinterpret_cast<const f32x4*>(fu + 4 * c4),
                                             *reinterpret_cast<const f32x4*>(fu + BN + 4 * c4));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y = rnd16<DT>(pre[j]);
                    if (h_seg == 0 && p.scale0 != 1.0f) y = rnd16<DT>(p.scale0 * y);
                    o[j] = (S)y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y = rnd16<DT>(v0[j] + v1[j] + (float)bv[j]);
                    if (h_seg == 0 && p.scale0 != 1.0f) y = rnd16<DT>(p.scale0 * y);
                    o[j] = (S)y;
                }
            }
            out_store(reinterpret_cast<V4e*>(h_dst + (((int64_t)bb * p.heads + hh) * p.n_pad + tok) * h_rs + d), o);
        } else if (GATE_RES) {
            const V4e gv = gpre[i], bv = bpre[i];
            f32x4 xv = xpre[i];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xv[j] = xv[j] + rnd16<DT>((float)gv[j] * rnd16<DT>(v0[j] + v1[j] + (float)bv[j]));
            out_store(reinterpret_cast<f32x4*>(p.x + (int64_t)(m0 + row) * pl_N + n0 + 4 * c4), xv);
            if constexpr (FOLD_P) {
                // the next site's operand and this unit's share of the row statistics; the partial sums go back into the unit's
                // own (dead) slot of the parking area
                const V4e sv = spre[i];
                const f32x2 cr = cpre[i];
                float s1 = 0.f, s2 = 0.f;
                V4e o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = xv[j] - cr[0];
                    s1 += d;
                    s2 = __builtin_fmaf(d, d, s2);
                    o[j] = (S)((d * cr[1]) * rnd16<DT>(1.0f + (float)sv[j]));
                }
                out_store(reinterpret_cast<V4e*>(p.ln_out + (int64_t)(m0 + row) * pl_N + n0 + 4 * c4), o);
                *reinterpret_cast<f32x2*>(red + row * RS + 4 * c4) = f32x2{s1, s2};
            }
        } else if constexpr (EPI == EPI_LINEAR) {
            epilogue_row4<DT, EPI>(p, m0 + row, n0 + 4 * c4, v0 + v1, bpre[i]);
        } else if constexpr (EPI == EPI_F32OUT) {
            // fp32 rows (p.x); the bias joins the rows from p.rows_per_batch on (epilogue_quad's EPI_F32OUT)
            const V4e bv = bpre[i];
            const bool wb = p.bias && m0 + row >= p.rows_per_batch;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = v0[j] + v1[j] + (wb ? (float)bv[j] : 0.f);
            out_store(reinterpret_cast<f32x4*>(p.x + (int64_t)(m0 + row) * pl_N + n0 + 4 * c4), o);
        } else if constexpr (EPI == EPI_LINEAR_FOLD) {
            out_store(reinterpret_cast<V4e*>(p.out + (int64_t)(m0 + row) * pl_N + n0 + 4 * c4),
                      fold_out4<DT>(p, fold_apply(v0 + v1, fstat[row], *reinterpret_cast<const f32x4*>(fu + 4 * c4),
                                                  *reinterpret_cast<const f32x4*>(fu + BN + 4 * c4))));
        }
    }
    if (pl_prof) pc_iss = __builtin_readcyclecounter();
    if constexpr (FOLD_P) {
        // row sums of the tile: 36 units per row -> thread (row, quarter) adds nine, the four quarters meet by DPP - a fixed order
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                 // H
        const int row = tid >> 2, qu = tid & 3;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(red + row * RS + 4 * (qu * 9 + k));
            s1 += v[0];
            s2 += v[1];
        }
        s1 = quad_sum(s1);
        s2 = quad_sum(s2);
        if (qu == 0 && m0 + row < pl_M)
            *reinterpret_cast<f32x2*>(p.fold_part + ((int64_t)(m0 + row) * nt + n0 / BN) * 2) = f32x2{s1, s2};
    }
    if constexpr (EPI == EPI_GATE_RESIDUAL_LN) {
        // ---------------- LayerNorm + modulate of the 128-row block, in the tail of the GEMM that completes its rows
        // (dit_crossattn.py:55-57: every gated residual add is followed by the LayerNorm of the next branch).  The nt column
        // tiles of a row block are nt workgroups; a row is complete when all of them have stored.  The waves of a workgroup meet
        // at a barrier, wave 0 counts the workgroup in on sync[2 g] and waits until all nt have arrived, a second barrier releases
        // the others, and every wave normalises ITS share of the block's rows - 128 / (8 nt) pairs, one row per half-wave, the row
        // body of ln_modulate_row32_kernel (ln_row.h: same bits).  Departures are counted on sync[2 g + 1]; the last one zeroes both
        // words, so a block's words are zero between launches whatever nt the next launch has.
        // MEASURED (round 4, same box, configs[1] step): bit-identical to the two launches and SLOWER in every protocol - one
        // arrival per wave 12.8 ms, one per workgroup 10.25 ms (agent-scope or same-XCD fences, L1 / L2 invalidate or none, sleep 2
        // or 8, counter read by load or by returning atomic: all within 0.03 ms), agent-scope release with its L2 write-back
        // 13.8 - 14.8 ms, against 8.99 ms with the LayerNorm as a launch of its own.  The fused kernel lasts 50 us where the GEMM
        // (28) and the LayerNorm (7) + a boundary (1.3) take 36: serialized same-address atomics cost ~0.5 us each and a waiting
        // workgroup sees the last arrival only microseconds later - a dependent kernel launch is the cheaper hand-over between CUs
        // on this part.  The route therefore runs only when the caller passes `sync` words (DiT.ln_in_kernel, off by default).
        // Why the in-kernel wait is safe: the waited-for workgroups never depend on the waiting ones and are dispatched no later
        // than them - the workgroups of a block have consecutive ids on one XCD (mt % 8 == 0, checked by the host; the id -> XCD
        // mapping itself by a probe launch, xcd_mapping_ok), the dispatcher hands out ids in order, so whenever the XCD's CUs are
        // all held by waiting workgroups the oldest block among them is complete (32 CUs >= 3 whole blocks of 8).  A bounded spin
        // (~1 s) turns a violated assumption into a counted error (primx_ln_sync_timeouts) instead of a hang.
        // p.ln_light (PRIMX_LN_MODE): bit 0 = same-XCD release (stores acknowledged by the shared L2, no L2 write-back), bits 1-2 =
        // acquire (0: agent-scope fence = L2 invalidate, 1: this CU's L1 only, 2: none), bit 3 = ONE arrival per workgroup (the
        // waves meet at a workgroup barrier, wave 0 counts in and waits, a second barrier releases the others) instead of one per
        // wave, bits 4-7 = s_sleep argument of the wait loop
        const int mode = p.ln_light;
        const bool per_wg = (mode & 8) != 0;
        const unsigned per = per_wg ? (unsigned)nt : (unsigned)nt * 8u;
        unsigned* cnt = p.sync + 2 * (m0 / BM);
        if (mode & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (per_wg) asm volatile("s_barrier" ::: "memory");                             // F: every wave's stores are out
        if (lane == 0 && (!per_wg || wave == 0)) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            const int nap = (mode >> 4) & 15;
            // (reading the counter with a returning atomic instead of the agent-scope load measured the same step time; an L1
            // invalidate + plain load never saw the arrivals: round 4, profiles/r4_experiments.txt)
            auto seen = [&]() -> unsigned { return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            while (seen() < per) {
                if (nap >= 8) __builtin_amdgcn_s_sleep(8);
                else if (nap >= 4) __builtin_amdgcn_s_sleep(4);
                else if (nap >= 2) __builtin_amdgcn_s_sleep(2);
                else __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 20)) {
                    atomicAdd(&g_ln_sync_timeouts, 1u);
                    break;
                }
            }
        }
        if (per_wg) asm volatile("s_barrier" ::: "memory");                             // G: the block's rows are complete
        const int acq = (mode >> 1) & 3;
        if (acq == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        else if (acq == 1) asm volatile("buffer_inv sc0" ::: "memory");
        else asm volatile("" ::: "memory");
        const int q = (n0 / BN) * 8 + wave;                   // this wave among the block's `per`
        for (int r = 2 * q + (lane >> 5); r < BM; r += 2 * (int)per) {
            const int m = m0 + r;
            if (m >= pl_M) continue;
            const int64_t bo = (int64_t)(m / p.rows_per_batch) * p.ln_mod_stride;
            ln_row32<DT, 9>(p.x + (int64_t)m * pl_N, p.ln_shift + bo, p.ln_scale + bo, p.ln_out + (int64_t)m * pl_N, lane & 31, p.ln_eps);
        }
        if (lane == 0 && (!per_wg || wave == 0)) {
            const unsigned d = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == per - 1) {                                // everybody has passed the wait: the words go back to zero
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (pl_prof) {
        __builtin_amdgcn_s_waitcnt(0);   // the stores have been acknowledged
        const unsigned long long pc3 = __builtin_readcyclecounter(), pr1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            atomicMin(&g_gemm_prof[0], pr0); atomicMax(&g_gemm_prof[1], pr1);
            atomicAdd(&g_gemm_prof[2], pc1 - pc0); atomicAdd(&g_gemm_prof[3], pc2 - pc1);
            atomicAdd(&g_gemm_prof[4], pc3 - pc2); atomicAdd(&g_gemm_prof[5], 1ull); atomicAdd(&g_gemm_prof[6], pr0);
            atomicAdd(&g_gemm_prof[7], ((pc_stg - pc2) << 32) | (pc_iss - pc_stg));          // (parking + the barrier behind it | the walk's issue)
            atomicAdd(&g_gemm_prof[8], pr1 - pr0); atomicAdd(&g_gemm_prof[9], pc3 - pc0);
            if (blockIdx.x < 4096) {
                unsigned xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                g_gemm_wg[blockIdx.x][0] = pr0; g_gemm_wg[blockIdx.x][1] = pr1 - pr0; g_gemm_wg[blockIdx.x][2] = pc3 - pc0;
                g_gemm_wg[blockIdx.x][3] = ((unsigned long long)(xcc & 15) << 32) | (unsigned)(pc3 - pc2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Big tile: 256 x 288, 8 waves = 4 (M) x 2 (N), each wave 64 x 144 = 4 x 9 tiles of 16x16x32.  Made for the GEMMs that the
// 128x144 kernel runs as several sequential rounds per CU (fc1: N = 4608 = 16 x 288 -> exactly 256 workgroups at M = 4096;
// qkv: 192; the batched to_k/to_v: 2464): every round pays ~5 us of un-overlapped prologue + epilogue (tools/gemm_ksweep.py),
// while here one barrier interval carries 4x the MFMAs per LDS byte and half the staged bytes per FLOP.
// Pipeline: the k dimension is staged in 32-wide slices through a 4-stage LDS-DMA ring (4 x 34,816 B) with ONE barrier per
// slice; the DMAs of slice h+3 are spread between the MFMAs of slice h and have two slices of MFMA time to land.  (The first
// form - a 2-stage ring of 64-wide tiles, two barriers per tile, DMAs issued behind the MFMAs - measured 754 + 562 cycles per
// k-tile of DMA wait + issue serial with the 2304 MFMA cycles and was removed in round 3.)
// Epilogue: MFMA operands swapped (accumulator = C^T: a lane owns one row and four consecutive columns), so the dense-output
// epilogues run straight from registers; the heads epilogue parks the tile in LDS and walks it in destination order.
// LDS rows are 64 bytes (4 chunks of 16 B); chunk' = chunk ^ 2*((row>>3)&1) makes the ds_read_b128 lane groups
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) hit 16 distinct 16-byte slots (rows r&3 pick the 64-byte quarter of a
// 256-byte bank window, the XOR separates rows 0-7 from 8-15 which the groups pair with chunk c and c+1).
// KT = 64 (round 6; K % 64 == 0): the operand stream in 128-BYTE row segments - whole cache lines - instead of 64-byte ones.  The way
// into a CU is bound by REQUESTS, not bytes (tools/probe/big_ingest.hip, profiles/r6_ingest.txt: this tile's stream alone, no MFMAs, takes
// 1530 cycles per 32-wide slice as 16 rows x 64 B per instruction and 980 as 8 rows x 128 B - the MFMAs need 1152; the shipped loop
// measured 1650: it ran at the request rate, not at the matrix pipe's).  The ring becomes TWO stages of 64-wide tiles (the same 139 KB;
// rows of 64 halves with the 128 x 144 kernels' chunk swizzle on the source address), ONE barrier per tile, placed two MFMA groups into
// the tile's second half - "everybody has read tile t, tile t + 1 has landed" - behind which tile t + 2 is requested into tile t's stage
// between the remaining MFMA groups; the first half of every tile runs without any barrier.
// LDS halves of one body (ring / heads staging area + the fold consumer's statistics): the __global__ wrappers own the block
template <int EPI, int KT>
constexpr int gemm288q_lds_halves() {
    constexpr int BM = 256, BN = 288, NST = KT == 64 ? 2 : 4, STAGE = (BM + BN) * KT, RS_ROWS = BN + 16, RS_VT = BM + 16;
    constexpr bool HEADS = EPI == EPI_HEADS || EPI == EPI_HEADS_FOLD, FOLD_C = EPI == EPI_HEADS_FOLD || EPI == EPI_LINEAR_FOLD;
    constexpr int STG = HEADS ? ((BM * RS_ROWS > BN * RS_VT) ? BM * RS_ROWS : BN * RS_VT) : 0;
    return ((NST * STAGE > STG) ? NST * STAGE : STG) + (FOLD_C ? BM * 4 + 2 * BN * 2 : 0);
}
// The kernel's body lives in gemm288q_body.inc (textual include: the pair kernel below instantiates it twice in one kernel)
template <int DT, int EPI, int KT = 32>
__global__ __launch_bounds__(512, 2) void gemm288q_dma_kernel(PRIMX_GEMM_PARAMS(DT)) {
    PRIMX_GEMM_ARGS(DT);
    __shared__ __attribute__((aligned(16))) typename T16<DT>::S smem[gemm288q_lds_halves<EPI, KT>()];
    const int bid = blockIdx.x;
#include "gemm288q_body.inc"
}

// Two problems in one launch (round 6): workgroups [0, n0) are the fold-consumer heads GEMM of the leading arguments (qkv of a DiT block:
// 192 workgroups at T = 4096 = 75 % of the CUs), workgroups [n0, grid) a plain heads GEMM `r` (the to_k / to_v projection of the NEXT
// block's conditioning tokens, 48 tiles) on CUs the first problem leaves idle.  Same bodies, same tiles, same bits as the two launches.
// n0 % 8 == 0, so that a rider's id keeps its XCD (id mod 8) for the tile walk.
template <int DT, int KT>
__global__ __launch_bounds__(512, 2) void gemm288q_pair_kernel(PRIMX_GEMM_PARAMS(DT), int n0, const GemmArgs<DT> r) {
    constexpr int H0 = gemm288q_lds_halves<EPI_HEADS_FOLD, KT>(), H1 = gemm288q_lds_halves<EPI_HEADS, KT>();
    __shared__ __attribute__((aligned(16))) typename T16<DT>::S smem[H0 > H1 ? H0 : H1];
    if ((int)blockIdx.x < n0) {
        constexpr int EPI = EPI_HEADS_FOLD;
        PRIMX_GEMM_ARGS(DT);
        const int bid = blockIdx.x;
#include "gemm288q_body.inc"
    } else {
        constexpr int EPI = EPI_HEADS;
        const GemmArgs<DT>& p = r;
        const typename T16<DT>::S *const pl_A = r.A, *const pl_W = r.W;
        const int pl_M = r.M, pl_N = r.N, pl_K = r.K, pl_xcd_gm = r.xcd_gm, pl_prof = 0, bid = (int)blockIdx.x - n0;
#include "gemm288q_body.inc"
    }
}

// ---------------------------------------------------------------------------------------------------
// (Rounds 3 - 5 had a TWO-PASS form of the big tile here for the Linear epilogue's one-round launches - gemm288p_dma_kernel: the 256 x 288
// tile as two 256 x 144 passes with loader waves, so that the first pass's stores drained under the second pass; fc1 + GELU at T = 4096
// 56 -> 50.5 us in the step.  It left in round 6: with the GELU arithmetic in packed fp32 (epilogue 19k -> 10k cycles), 128-byte row
// segments (KT = 64) and the carried weight prefetch in gemm288q_dma_kernel, the one-pass kernel runs the same launch in 46.3 us and
// the fc2 launch behind it as fast as before - profiles/r6_fc1_onepass.txt.  The two-pass loop read 37 LDS bytes per kFLOP against 22.)

// ---------------------------------------------------------------------------------------------------
// (Round 5's persistent-pass kernel gemm144pp_dma_kernel - the two-pass tile's 256 x 144 pass as one workgroup per CU walking a list of
// passes, opt-in, measured a tie at T = 32768 (profiles/r5_pp_experiments.txt) - left the library in round 6.  The round-6 probes say why it
// could only tie: a 256 x 144 pass needs 44 bytes per clock and CU of operands at the MFMA rate, and the way into a CU takes 35 - 48 of
// them as 128-byte requests whoever issues them (profiles/r6_largeM_diagnosis.txt); the seams it removed were never the bound.)

// (Round 4 built a PERSISTENT form of the big tile here - four waves of 128 x 144 with the whole register file, one per SIMD, walking
// many tiles as one operand stream, first fed by LDS-DMA, then through registers - to overlap ring fill and store drain at T >= 8192,
// where every CU runs eight tiles back to back.  Both forms were correct and both lost to the 8-wave kernel above (fc1 at T = 32768:
// 583 / 620 vs 453 us; the batch-8 step 66.8 / 70.4 vs 62.7 ms): with one wave per SIMD nobody covers the wave while it sits in the
// DMA unit's queue or waits for its staged slice, and nobody covers it in the epilogue.  Removed; measurements, timelines and the
// inline-asm-MFMA hazard found on the way: profiles/r4_experiments.txt sections 2 - 3, DESIGN_LOG.md section 10.)

static const bool g_no_big = [] {   // PRIMX_GEMM_NOBIG=1 disables the 256x288 tile (A/B measurements)
    const char* e = getenv("PRIMX_GEMM_NOBIG");
    return e && e[0] == '1';
}();

static const bool g_loader = [] {   // PRIMX_GEMM_LOADER=0: the 128x144 kernel without loader waves (gemm144_dma_kernel) everywhere
    const char* e = getenv("PRIMX_GEMM_LOADER");
    return !(e && e[0] == '0');
}();

static const int g_big_min = [] {   // PRIMX_GEMM_BIG_MIN: fewest 256x288 workgroups for which the dense-output epilogues take the big tile
    const char* e = getenv("PRIMX_GEMM_BIG_MIN");   // (224: one launch fills the chip; 112 when two half-batch streams run side by side)
    return e ? atoi(e) : 224;
}();

static const int g_big_heads_min = [] {   // fewest 256x288 workgroups for which the heads epilogue takes the big tile
    const char* e = getenv("PRIMX_GEMM_BIGHEADS_MIN");
    return e ? atoi(e) : 160;
}();

static const bool g_xcd2d = [] {   // PRIMX_GEMM_XCD2D=0: whole tile rows per XCD in the 256x288 kernel (A/B measurements)
    const char* e = getenv("PRIMX_GEMM_XCD2D");
    return !(e && e[0] == '0');
}();

// LayerNorm in the tail of the gate-residual GEMM (primx_linear_gate_residual_ln): PRIMX_LN_FUSE=0 always takes the two-launch
// route; PRIMX_LN_MODE selects the fences and the arrival protocol (A/B measurements);
// PRIMX_LN_FUSE_MAXGRID bounds the grid of a fused launch.
static const bool g_ln_fuse = [] {
    const char* e = getenv("PRIMX_LN_FUSE");
    return !(e && e[0] == '0');
}();
static const int g_ln_mode = [] {   // PRIMX_LN_MODE: see the tail of gemm144l_dma_kernel (bit 0 light release | acquire kind << 1 | per-workgroup arrival 8 | sleep << 4)
    const char* e = getenv("PRIMX_LN_MODE");
    return e ? atoi(e) : (1 | (0 << 1) | 8 | (2 << 4));   // 41: same-XCD release (stores acknowledged by the shared L2), agent-scope acquire, one arrival per workgroup
}();
static const int g_ln_maxgrid = [] {
    const char* e = getenv("PRIMX_LN_FUSE_MAXGRID");
    return e ? atoi(e) : 2048;
}();
// The fused route's same-XCD fences rest on the workgroup -> XCD mapping that xcd_remap assumes everywhere (workgroup id mod 8 =
// XCD): checked ONCE per device by a probe launch (every workgroup reports the XCC it runs on) before the first fused launch; a
// device that maps differently (another partition mode, fewer XCDs) takes the two-launch route.
__device__ unsigned g_xcd_probe[256];
__global__ void xcd_probe_kernel() {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) g_xcd_probe[blockIdx.x] = xcc & 15;
}
static bool xcd_mapping_ok() {
    static int state[64] = {0};   // per device: 0 unknown, 1 ok, -1 not
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (state[dev] == 0) {
        unsigned h[256];
        hipLaunchKernelGGL(xcd_probe_kernel, dim3(256), dim3(64), 0, 0);
        bool ok = hipMemcpyFromSymbol(h, HIP_SYMBOL(g_xcd_probe), sizeof(h)) == hipSuccess;
        for (int i = 0; ok && i < 256; ++i) ok = h[i] == h[i & 7];
        for (int i = 0; ok && i < 8; ++i)
            for (int j = 0; j < i; ++j) ok = ok && h[i] != h[j];
        state[dev] = ok ? 1 : -1;
    }
    return state[dev] == 1;
}
thread_local bool g_ln_fused = false;   // did the last launch144_dma on this thread run the LayerNorm in the GEMM's tail?

static const bool g_gemm_prof_on = [] {   // PRIMX_GEMM_PROF=1: synchronous launches + per-workgroup timeline print (8-wave kernels)
    const char* e = getenv("PRIMX_GEMM_PROF");
    return e && atoi(e) != 0;
}();

// The kernel instantiation the last GEMM entry point called on this thread selected, spelled as rocprofv3 prints it
// (primx_last_gemm_kernel(), include/primx_hip.h): bench.py tags its per-launch timings with what the C side actually
// launched instead of a Python restatement of the dispatch rules below.
thread_local char g_last_gemm_kernel[112] = "";
#define PRIMX_NOTE_KERNEL(...) snprintf(g_last_gemm_kernel, sizeof(g_last_gemm_kernel), __VA_ARGS__)

// XCD block shape gm (8 / gm column groups) + the sub-block an XCD's 32 CUs work on together (xcd_tile2d): sr = the largest divisor
// of the block's bm rows that is <= 8 (bm itself when bm <= 8), sc = 32 / sr columns (at most the block's bn).
static int xcd_pack(int gm, int mtb, int ntb, bool row_major) {
    const int bm = mtb / gm, bn = ntb / (8 / gm);
    if (row_major) return gm | (1 << 8) | (bn << 16);      // sr = 1, sc = bn: n fastest over the whole block
    int sr = bm;
    if (bm > 8)
        for (int d = 8; d >= 1; --d)
            if (bm % d == 0) { sr = d; break; }
    const int sc = std::max(1, std::min(bn, 32 / sr));
    return gm | (sr << 8) | (sc << 16);
}


static const bool g_kt32 = [] {   // PRIMX_GEMM_KT32=1: the 256x288 kernel on its round 1 - 5 ring of 32-wide slices (64-byte requests) for every K
    const char* e = getenv("PRIMX_GEMM_KT32");
    return e && atoi(e) != 0;
}();

static const int g_kt64_min = [] {   // PRIMX_GEMM_KT64_MIN: fewest 256x288 workgroups for which the dense-output epilogues take 128-byte row segments
    const char* e = getenv("PRIMX_GEMM_KT64_MIN");
    return e ? atoi(e) : -1;   // (-1: the rule of launch288q)
}();

// Which ring: 128-byte row segments (KT = 64) for the dense-output epilogues of launches with more than one round of workgroups -
// measured (profiles/r6_kt64_experiments.txt, same box): fc1 + GELU at T = 32768 466 -> 437 us, proj 145 -> 134, fc2 371 -> 345; at
// ONE round (qkv at T = 4096: 192 workgroups) the two-stage ring's longer first wait costs more than the request rate gives (44.5 ->
// 46.5 us), and the heads epilogues neither gain (T = 16384 qkv 137 vs 137 us) nor fit the registers (the compiler spills 7 - 29
// dwords next to their two main loops): both keep the 32-wide ring.  (KT = 64 addresses its operands by 32-bit byte offsets.)
static const bool g_heads_kt64 = [] {   // PRIMX_GEMM_HEADS_KT32=1: the heads epilogues of the 256x288 kernel on the ring of 32-wide slices (rounds 1 - 6a)
    const char* e = getenv("PRIMX_GEMM_HEADS_KT32");
    return !(e && atoi(e) != 0);
}();

// The heads epilogues on the 128-byte ring (end of round 6): with the rolling fragment window of gemm288q_body.inc they fit the registers
// (no scratch), and their k-loop leaves the request-bound regime like the dense epilogues' did.
template <int DT>
static bool heads_kt64(const GemmArgs<DT>& x) {
    return g_heads_kt64 && !g_kt32 && x.K % 64 == 0 && x.K >= 128 && (int64_t)x.M * x.K < (1ll << 31) && (int64_t)x.N * x.K < (1ll << 31);
}

template <int DT, int EPI>
static void launch288q(const GemmArgs<DT>& x, dim3 grid, hipStream_t st) {
    constexpr bool DENSE = EPI == EPI_LINEAR || EPI == EPI_GATE_RESIDUAL || EPI == EPI_GATE_RESIDUAL_FOLD || EPI == EPI_LINEAR_FOLD ||
                           EPI == EPI_RES;
    if constexpr (DENSE) {
        const bool full = EPI != EPI_GATE_RESIDUAL || (x.M % 256 == 0 && x.rows_per_batch % 256 == 0);   // (its pipelined epilogue: no ragged tile, one gate row per tile)
        // (default: the Linear epilogues always - fc1 at T = 4096, ONE round of 256 workgroups, 50.5 -> 45.0 us in the step against the
        //  two-pass kernel once this kernel carries the weight prefetch; the read-modify-write epilogues from two rounds on)
        const int kt64_min = g_kt64_min >= 0 ? g_kt64_min : (EPI == EPI_LINEAR || EPI == EPI_LINEAR_FOLD) ? 1 : 257;
        if (x.K % 64 == 0 && !g_kt32 && full && (int)grid.x >= kt64_min && (int64_t)x.M * x.K < (1ll << 31) && (int64_t)x.N * x.K < (1ll << 31)) {
            PRIMX_NOTE_KERNEL("gemm288q_dma_kernel<%d, %d, 64>", DT, EPI);
            hipLaunchKernelGGL((gemm288q_dma_kernel<DT, EPI, 64>), grid, dim3(512), 0, st, PRIMX_GEMM_PASS(x));
            return;
        }
    }
    if constexpr (EPI == EPI_HEADS || EPI == EPI_HEADS_FOLD) {
        if (heads_kt64<DT>(x)) {
            PRIMX_NOTE_KERNEL("gemm288q_dma_kernel<%d, %d, 64>", DT, EPI);
            hipLaunchKernelGGL((gemm288q_dma_kernel<DT, EPI, 64>), grid, dim3(512), 0, st, PRIMX_GEMM_PASS(x));
            return;
        }
    }
    {
        PRIMX_NOTE_KERNEL("gemm288q_dma_kernel<%d, %d, 32>", DT, EPI);
        hipLaunchKernelGGL((gemm288q_dma_kernel<DT, EPI, 32>), grid, dim3(512), 0, st, PRIMX_GEMM_PASS(x));
    }
}

// XCD block shape of a 256 x 288 launch (packed for xcd_tile2d; 0 = whole tile rows per XCD): minimise (A bytes x column groups + W bytes
// x row groups) over the splits the tile grid allows
template <int DT>
static int big_xcd_gm(const GemmArgs<DT>& a, bool heads) {
    if (!g_xcd2d) return a.xcd_gm;
    const int mtb = (a.M + 255) / 256, ntb = a.N / 288;
    double best = 1e300;
    int g = a.xcd_gm;
    for (int gm = 1; gm <= 8; gm *= 2) {
        const int gn = 8 / gm;
        if (mtb % gm || ntb % gn) continue;
        const double cost = (double)a.M * gn + (double)a.N * gm;      // x K x 2 bytes each
        if (cost < best) { best = cost; g = gm; }
    }
    // (the heads epilogue keeps the row-major walk: the batched K / V projection measured 210 vs 217 us with sub-blocks - its rounds
    // are bound by the two-layout scatter of the tile, not by operand traffic - while the dense-output GEMMs of a large batch
    // gain: fc1 at T = 32768 440 -> 410 us, the batch-8 step 61.25 -> 60.86 ms)
    return g > 0 ? xcd_pack(g, mtb, ntb, heads) : g;
}

template <int DT, int EPI, int BIG = 0>
void launch144_dma(const GemmArgs<DT>& a, int mt, hipStream_t st) {
    const dim3 grid(BIG ? ((a.M + 255) / 256) * (a.N / 288) : mt * (a.N / 144));
    GemmArgs<DT> a2 = a;
    a2.ln_light = g_ln_mode;
    if (BIG) a2.xcd_gm = big_xcd_gm<DT>(a, EPI == EPI_HEADS || EPI == EPI_HEADS_FOLD);
    constexpr bool FOLD_EPI = EPI == EPI_GATE_RESIDUAL_FOLD || EPI == EPI_HEADS_FOLD || EPI == EPI_LINEAR_FOLD;
    // loader-wave kernel: the row-major epilogues (heads: token-major segments whose tiles stay inside one segment)
    bool loader_ok = !BIG && (EPI == EPI_LINEAR || EPI == EPI_GATE_RESIDUAL || EPI == EPI_GATE_RESIDUAL_FOLD || EPI == EPI_LINEAR_FOLD);
    if (!BIG && (EPI == EPI_HEADS || EPI == EPI_HEADS_FOLD) && a.heads > 0) {
        const int per = a.heads * a.dh;
        loader_ok = per % 144 == 0 && a.dh >= 48 && a.dh % 4 == 0 && a.rows_per_batch >= 128;
        for (int sgi = 0; sgi < a.n_seg; ++sgi) loader_ok = loader_ok && a.kind[sgi] != PRIMX_HEADS_VT;
    }
    auto go = [&](const GemmArgs<DT>& x) {
        // the LayerNorm-fold epilogues exist in exactly one kernel per tile shape (launch_fold checked the shape)
        if constexpr (FOLD_EPI) {
            if constexpr (BIG) {
                launch288q<DT, EPI>(x, grid, st);
            } else {
                PRIMX_NOTE_KERNEL("gemm144l_dma_kernel<%d, %d>", DT, EPI);
                hipLaunchKernelGGL((gemm144l_dma_kernel<DT, EPI>), grid, dim3(640), 0, st, PRIMX_GEMM_PASS(x));
            }
            return;
        } else {
        if (BIG) {
            launch288q<DT, EPI>(x, grid, st);
        } else if (g_loader && loader_ok) {
            if constexpr (EPI == EPI_GATE_RESIDUAL) {
                // LayerNorm of the updated rows in the kernel's tail: N = 1152 (nine 128-column chunks per half-wave row), 8-byte
                // aligned modulation vectors, and every row block's column tiles on ONE XCD with consecutive ids (mt % 8 == 0:
                // xcd_remap hands each XCD mt / 8 whole blocks) - see the kernel for why the in-kernel wait is safe then
                const int nt = x.N / 144;
                if (x.ln_out && x.sync && g_ln_fuse && x.N == 1152 && mt % 8 == 0 && (int)grid.x <= g_ln_maxgrid && nt * 8 <= 128 &&
                    (((uintptr_t)x.ln_shift | (uintptr_t)x.ln_scale | (uintptr_t)x.ln_out) & 7) == 0 && x.ln_mod_stride % 4 == 0 &&
                    xcd_mapping_ok()) {
                    PRIMX_NOTE_KERNEL("gemm144l_dma_kernel<%d, %d>", DT, EPI_GATE_RESIDUAL_LN);
                    hipLaunchKernelGGL((gemm144l_dma_kernel<DT, EPI_GATE_RESIDUAL_LN>), grid, dim3(640), 0, st, PRIMX_GEMM_PASS(x));
                    g_ln_fused = true;
                    return;
                }
            }
            if constexpr (EPI == EPI_LINEAR || EPI == EPI_GATE_RESIDUAL || EPI == EPI_HEADS) {
                PRIMX_NOTE_KERNEL("gemm144l_dma_kernel<%d, %d>", DT, EPI);
                hipLaunchKernelGGL((gemm144l_dma_kernel<DT, EPI>), grid, dim3(640), 0, st, PRIMX_GEMM_PASS(x));
            }
        } else {
            PRIMX_NOTE_KERNEL("gemm144_dma_kernel<%d, %d>", DT, EPI);
            hipLaunchKernelGGL((gemm144_dma_kernel<DT, EPI>), grid, dim3(512), 0, st, PRIMX_GEMM_PASS(x));
        }
        }
    };
    if (!g_gemm_prof_on) {
        go(a2);
        return;
    }
    GemmArgs<DT> b = a2;
    b.prof = 1;
    unsigned long long z[12] = {~0ull, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, r[12];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), z, sizeof(z));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    go(b);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_gemm_prof), sizeof(r));
    const double n = r[5] ? (double)r[5] : 1.0;
    fprintf(stderr, "%s<%d,%d> M=%d N=%d K=%d: %llu workgroups, events %.1f us, first start -> last end %.1f us, mean start offset "
                    "%.1f us, shader clock %.2f GHz (core cycles / 100 MHz ticks per workgroup); per workgroup (core cycles): entry->tile0 %.0f | main loop "
                    "%.0f | epilogue %.0f (LDS staging %.0f, read+store issue %.0f)\n",
            BIG ? "gemm288q_dma"
                : (FOLD_EPI || (g_loader && loader_ok)) ? "gemm144l_dma" : "gemm144_dma", DT, EPI, a.M, a.N, a.K,
            r[5], ms * 1e3, (r[1] - r[0]) * 0.01, (r[6] / n - (double)r[0]) * 0.01, r[8] ? (double)r[9] / (double)r[8] * 0.1 : 0.0,
            r[2] / n, r[3] / n, r[4] / n, (r[7] >> 32) / n,
            (r[7] & 0xffffffffull) / n);
    {   // distribution over the workgroups: the kernel is as long as its slowest one
        static unsigned long long wg[4096][4];
        const int nw = (int)(r[5] < 4096 ? r[5] : 4096);
        (void)hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_gemm_wg), sizeof(unsigned long long) * 4 * nw);
        double life[4096], xs[8] = {0}, xe[8] = {0}, xend[8] = {0};
        int xn[8] = {0};
        for (int i = 0; i < nw; ++i) {
            life[i] = wg[i][1] * 0.01;
            const int x = (int)(wg[i][3] >> 32) & 7;
            xs[x] += life[i]; xe[x] += (double)(wg[i][3] & 0xffffffffull); ++xn[x];
            const double end = (wg[i][0] + wg[i][1] - r[0]) * 0.01;
            if (end > xend[x]) xend[x] = end;
        }
        double srt[4096];
        for (int i = 0; i < nw; ++i) srt[i] = life[i];
        for (int i = 1; i < nw; ++i) { double v = srt[i]; int j = i; while (j > 0 && srt[j - 1] > v) { srt[j] = srt[j - 1]; --j; } srt[j] = v; }
        fprintf(stderr, "   workgroup lifetimes (us): min %.1f | median %.1f | p90 %.1f | max %.1f;  per XCD mean lifetime / mean epilogue cycles / last end:",
                srt[0], srt[nw / 2], srt[nw * 9 / 10], srt[nw - 1]);
        for (int x = 0; x < 8; ++x) if (xn[x]) fprintf(stderr, "  [%d] %.1f / %.0f / %.1f", x, xs[x] / xn[x], xe[x] / xn[x], xend[x]);
        fprintf(stderr, "\n");
        if (nw > 256) {   // rounds of 256 workgroups in start order: when does a CU's next workgroup begin after the previous one ended?
            static int ord[4096];
            for (int i = 0; i < nw; ++i) ord[i] = i;
            for (int i = 1; i < nw; ++i) { const int v = ord[i]; int j = i; while (j > 0 && wg[ord[j - 1]][0] > wg[v][0]) { ord[j] = ord[j - 1]; --j; } ord[j] = v; }
            fprintf(stderr, "   rounds of 256 in start order (us from the first start): mean start / mean end:");
            for (int r0 = 0; r0 < nw && r0 < 256 * 10; r0 += 256) {
                double ms_ = 0, me_ = 0;
                const int n_ = nw - r0 < 256 ? nw - r0 : 256;
                for (int i = r0; i < r0 + n_; ++i) { ms_ += (wg[ord[i]][0] - r[0]) * 0.01; me_ += (wg[ord[i]][0] + wg[ord[i]][1] - r[0]) * 0.01; }
                fprintf(stderr, "  %.1f / %.1f", ms_ / n_, me_ / n_);
            }
            fprintf(stderr, "\n");
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

template <int DT, int EPI, int GATHER = 0>
int launch(const GemmArgs<DT>& a_in, hipStream_t st, const char* name) {
    GemmArgs<DT> a = a_in;
    PRIMX_REQUIRE(a.A && a.W, "%s: null operand", name);
    PRIMX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 8 == 0, "%s: need M,N>0 and K %% 8 == 0 (M=%d N=%d K=%d)",
                  name, a.M, a.N, a.K);
    const int mt = (a.M + 127) / 128;
    const bool tail = (a.K % BK) != 0;
    // 256x288 tile: only where same-box A/B showed a win - the dense-output epilogues with enough workgroups to fill
    // the chip (fc1 at T = 4096: exactly 256; every large-batch GEMM).  The heads epilogue stays on the 128x144 kernel
    // (qkv would be 192 workgroups = 75 % of the CUs, and the scatter epilogue spilled at this tile's register budget).
    bool use_big = !g_no_big && a.N % 288 == 0 && ((a.M + 255) / 256) * (a.N / 288) >= g_big_min &&
                   (EPI == EPI_LINEAR || EPI == EPI_RES || EPI == EPI_GATE_RESIDUAL);
    // Heads epilogue on the 256x288 tile (LDS-staged scatter, gemm288q only): tiles must cover whole heads of one
    // segment and one batch entry.  It already pays at 192 workgroups (qkv at T = 4096: 75 % of the CUs, one round
    // instead of three rounds of 128x144 tiles); PRIMX_GEMM_BIGHEADS_MIN moves the threshold (0 = never).
    if (EPI == EPI_HEADS && !g_no_big && g_big_heads_min > 0 && a.heads > 0) {
        const int per = a.heads * a.dh;
        // (dh >= 32: the epilogue finds the head of a column with at most 8 compare-subtract steps, 288 / dh <= 9 heads per tile)
        use_big = a.N % 288 == 0 && per % 288 == 0 && 288 % a.dh == 0 && a.dh % 8 == 0 && a.dh >= 32 && a.rows_per_batch % 256 == 0 &&
                  (a.M / 256) * (a.N / 288) >= g_big_heads_min;
    }
    // (the register-staged kernels check every 16-byte chunk against K: one compare per load buys half the instantiations of a
    // compile-time "K has a tail" flag; the LDS-DMA kernels need K % 64 == 0.  The narrow 128 x 32 tile exists for the epilogues
    // that meet N <= 32 - the VAE's convolutions and plain Linear)
    if (a.N <= 32 && (EPI == EPI_RES || EPI == EPI_LINEAR)) {
        if constexpr (EPI == EPI_RES || EPI == EPI_LINEAR) {
            PRIMX_NOTE_KERNEL("gemm_kernel<%d, %d, 32, 4, 1, 1, 1, %d>", DT, EPI, GATHER);
            hipLaunchKernelGGL((gemm_kernel<DT, EPI, 32, 4, 1, 1, 1, GATHER>), dim3(mt * ((a.N + 31) / 32)), dim3(256), 0, st, a);
        }
    } else if (use_big && !tail && !GATHER) {
        launch144_dma<DT, EPI, 1>(a, mt, st);
    } else if (a.N % 144 == 0 && !GATHER && !tail) {
        launch144_dma<DT, EPI>(a, mt, st);
    } else {
        PRIMX_NOTE_KERNEL("gemm_kernel<%d, %d, 32, 2, 2, 2, 2, %d>", DT, EPI, GATHER);
        hipLaunchKernelGGL((gemm_kernel<DT, EPI, 32, 2, 2, 2, 2, GATHER>), dim3(mt * ((a.N + 127) / 128)), dim3(256), 0, st, a);
    }
    PRIMX_CHECK_LAUNCH(name);
    return PRIMX_OK;
}

// The LayerNorm-fold GEMMs (see fold_stats_load): explicit kernel choice, an error where the fold kernels do not cover the shape - the
// caller decides per model whether it folds (DiT: ops.fold_supported) and keeps the LayerNorm launches otherwise.
template <int DT, int EPI>
int launch_fold(const GemmArgs<DT>& a, hipStream_t st, const char* name) {
    PRIMX_REQUIRE(a.A && a.W, "%s: null operand", name);
    PRIMX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % BK == 0 && a.N % 144 == 0,
                  "%s: the fold kernels need N %% 144 == 0 and K %% 64 == 0 (M=%d N=%d K=%d)", name, a.M, a.N, a.K);
    const int mt = (a.M + 127) / 128;
    if constexpr (EPI == EPI_GATE_RESIDUAL_FOLD) {
        PRIMX_REQUIRE(a.N / 144 <= 8, "%s: at most 8 column tiles (N <= 1152), the consumers read 8 partial sums per row (N=%d)", name, a.N);
        PRIMX_REQUIRE((((uintptr_t)a.ln_scale | (uintptr_t)a.ln_out | (uintptr_t)a.fold_part | (uintptr_t)a.fold_c) & 7) == 0 &&
                          a.ln_mod_stride % 4 == 0,
                      "%s: the scale vectors, the operand, the partial sums and the (centre, scale) pairs must be 8-byte aligned", name);
        // the 256 x 288 tile where the unfolded gate-residual GEMM takes it (launch<>: a large batch)
        if (!g_no_big && a.N % 288 == 0 && ((a.M + 255) / 256) * (a.N / 288) >= g_big_min) launch144_dma<DT, EPI, 1>(a, mt, st);
        else launch144_dma<DT, EPI>(a, mt, st);
    } else {
        PRIMX_REQUIRE(a.K % 144 == 0 && a.fold_parts == a.K / 144 && a.fold_parts <= 8,
                      "%s: K must be the producer's N: a multiple of 144, at most 1152 (K=%d)", name, a.K);
        PRIMX_REQUIRE((((uintptr_t)a.fold_u | (uintptr_t)a.fold_v) & 15) == 0 &&
                          (((uintptr_t)a.fold_part | (uintptr_t)a.fold_c | (uintptr_t)a.fold_c_out) & 7) == 0,
                      "%s: u / v must be 16-byte aligned, the partial sums and the (centre, scale) pairs 8-byte aligned", name);
        if constexpr (EPI == EPI_HEADS_FOLD) {
            const int per = a.heads * a.dh;
            const bool big = !g_no_big && g_big_heads_min > 0 && a.N % 288 == 0 && per % 288 == 0 && 288 % a.dh == 0 && a.dh % 8 == 0 &&
                             a.dh >= 32 && a.rows_per_batch % 256 == 0 && (a.M / 256) * (a.N / 288) >= g_big_heads_min;
            if (big) {
                launch144_dma<DT, EPI, 1>(a, mt, st);
            } else {
                bool ok = per % 144 == 0 && a.dh >= 48 && a.dh % 4 == 0 && a.rows_per_batch >= 128 && g_loader;
                for (int sgi = 0; sgi < a.n_seg; ++sgi) ok = ok && a.kind[sgi] != PRIMX_HEADS_VT;
                PRIMX_REQUIRE(ok, "%s: head layout outside the fold kernels (heads=%d dh=%d rows_per_batch=%d)", name, a.heads, a.dh,
                              a.rows_per_batch);
                launch144_dma<DT, EPI>(a, mt, st);
            }
        } else {
            const int wgs = ((a.M + 255) / 256) * (a.N / 288);
            if (!g_no_big && a.N % 288 == 0 && wgs >= g_big_min) launch144_dma<DT, EPI, 1>(a, mt, st);
            else launch144_dma<DT, EPI>(a, mt, st);
        }
    }
    PRIMX_CHECK_LAUNCH(name);
    return PRIMX_OK;
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// Few-row Linear (M <= 8: the adaLN modulation of every block from the timestep embeddings - B_e rows per step, or eight
// timesteps of a planned sampling loop per pass, DiT.plan_timesteps; one 674 MB weight stream per pass at DiT-XL).  HBM-bound:
// W rows stream with non-temporal 16-byte loads, A (a few KB) comes from L1, M x 4 fp32 partial sums per lane.  The MFMA tiles
// spent 180 us on it (256-row tiles with 2 valid rows: 1.26 GB through the L2 -> LDS path); PRIMX_GEMM_NOGEMV=1 goes back to them.
// ROWS = 4 or 8: a row's arithmetic (k order per lane, 16-lane reduction) does not depend on ROWS or on M, so a row computed
// alone, in a batch of 4 or in a batch of 8 is bit-identical (the per-loop modulation table of DiT.plan_timesteps relies on it).
// A wave is four 16-lane groups; a group owns GEMV_COLS = 4 columns and its lanes split K in 16-byte chunks (K = 1152: nine
// chunks per lane, every lane busy; the first form - 64 lanes per 4 columns - ran 2.25 iterations on 3 and then spent as many
// instructions on 6-step wave reductions of its M x 4 sums as on the products: 310 us for 8 rows).  The reduction is four DPP
// adds inside the group.
constexpr int GEMV_MAX_ROWS16 = 8, GEMV_COLS = 4;
__device__ __forceinline__ float group16_sum(float v) {
    // xor-butterfly over the 16 lanes of a DPP row: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
    return v;
}

template <int DT, int GEMV_ROWS>
__global__ __launch_bounds__(256) void gemv16_kernel(const typename T16<DT>::S* __restrict__ A,
                                                     const typename T16<DT>::S* __restrict__ W,
                                                     const typename T16<DT>::S* __restrict__ bias,
                                                     typename T16<DT>::S* __restrict__ out, int M, int N, int K) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    using V4 = typename T16<DT>::V4;
    const int lane = threadIdx.x & 63, l16 = lane & 15;
    const int n0 = (((blockIdx.x * 4 + (threadIdx.x >> 6)) << 2) + (lane >> 4)) * GEMV_COLS;
    const bool live = n0 < N;                 // N % GEMV_COLS == 0: a group's columns are all inside or all outside
    const int nc = live ? n0 : 0;             // dead groups stream valid addresses and store nothing
    float acc[GEMV_ROWS][GEMV_COLS] = {};
    for (int kc = l16; kc < K / 8; kc += 16) {
        V8 w[GEMV_COLS];
#pragma unroll
        for (int c = 0; c < GEMV_COLS; ++c)   // streamed once per forward: non-temporal
            w[c] = __builtin_nontemporal_load(reinterpret_cast<const V8*>(W + (int64_t)(nc + c) * K + kc * 8));
        // rows beyond M re-read row M - 1 (results discarded): no branch in the loop, so all loads of an iteration are in
        // flight together
        V8 a[GEMV_ROWS];
#pragma unroll
        for (int m = 0; m < GEMV_ROWS; ++m) a[m] = *reinterpret_cast<const V8*>(A + (int64_t)min(m, M - 1) * K + kc * 8);
#pragma unroll
        for (int m = 0; m < GEMV_ROWS; ++m) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float af = (float)a[m][e];
#pragma unroll
                for (int c = 0; c < GEMV_COLS; ++c) acc[m][c] = fmaf(af, (float)w[c][e], acc[m][c]);
            }
        }
    }
    float bf[GEMV_COLS];
#pragma unroll
    for (int c = 0; c < GEMV_COLS; ++c) bf[c] = bias ? (float)bias[nc + c] : 0.f;
#pragma unroll
    for (int m = 0; m < GEMV_ROWS; ++m) {
        V4 o;
#pragma unroll
        for (int c = 0; c < GEMV_COLS; ++c) o[c] = (S)(group16_sum(acc[m][c]) + bf[c]);
        if (m < M && live && l16 == 0) *reinterpret_cast<V4*>(out + (int64_t)m * N + n0) = o;
    }
}

static const bool g_no_gemv = [] {
    const char* e = getenv("PRIMX_GEMM_NOGEMV");
    return e && e[0] == '1';
}();

}  // namespace

extern "C" const char* primx_last_gemm_kernel(void) { return g_last_gemm_kernel; }

extern "C" int primx_ln_sync_timeouts(void) {
    unsigned v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ln_sync_timeouts), sizeof(v));   // (synchronises with the device)
    return (int)v;
}

extern "C" int primx_linear(const void* A, const void* W, const void* bias, void* out, int M, int N, int K, int dtype,
                            int act, float out_scale, const void* prefetch, int64_t prefetch_bytes, void* stream) {
    PRIMX_REQUIRE(out, "primx_linear: null output");
    PRIMX_REQUIRE(act == PRIMX_ACT_NONE || act == PRIMX_ACT_GELU_TANH || act == PRIMX_ACT_GELU_ERF, "primx_linear: bad activation code");
    PRIMX_REQUIRE((prefetch != nullptr) == (prefetch_bytes > 0) && prefetch_bytes >= 0,
                  "primx_linear: the prefetch range is (pointer, bytes > 0) or (NULL, 0)");   // (the few-row kernel below ignores it)
    if (M > 0 && M <= GEMV_MAX_ROWS16 && N % GEMV_COLS == 0 && K > 0 && K % 8 == 0 && act == PRIMX_ACT_NONE &&
        out_scale == 1.0f && !g_no_gemv) {
        PRIMX_REQUIRE(A && W, "primx_linear: null operand");
        PRIMX_DISPATCH_16(dtype, "primx_linear", {
            using S = typename T16<DT>::S;
            const dim3 grid((N / GEMV_COLS + 15) / 16);   // 4 waves x 4 groups of 4 columns per workgroup
            PRIMX_NOTE_KERNEL("gemv16_kernel<%d, %d>", DT, M <= 4 ? 4 : 8);
            if (M <= 4)
                hipLaunchKernelGGL((gemv16_kernel<DT, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const S*)A, (const S*)W,
                                   (const S*)bias, (S*)out, M, N, K);
            else
                hipLaunchKernelGGL((gemv16_kernel<DT, 8>), grid, dim3(256), 0, (hipStream_t)stream, (const S*)A, (const S*)W,
                                   (const S*)bias, (S*)out, M, N, K);
            PRIMX_CHECK_LAUNCH("primx_linear");
            return PRIMX_OK;
        });
    }
    PRIMX_DISPATCH_16(dtype, "primx_linear", {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W; a.bias = (const S*)bias;
        a.M = M; a.N = N; a.K = K;
        a.out = (S*)out; a.act = act; a.out_scale = out_scale;
        if (int rc = set_prefetch<DT>(a, prefetch, prefetch_bytes, "primx_linear")) return rc;
        return launch<DT, EPI_LINEAR>(a, (hipStream_t)stream, "primx_linear");
    });
    return PRIMX_OK;
}

extern "C" int primx_linear_residual(const void* A, const void* W, const void* bias, const void* res, float scale,
                                     void* out, int M, int N, int K, int dtype, void* stream) {
    PRIMX_REQUIRE(out, "primx_linear_residual: null output");
    PRIMX_DISPATCH_16(dtype, "primx_linear_residual", {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W; a.bias = (const S*)bias;
        a.M = M; a.N = N; a.K = K;
        a.out = (S*)out; a.res = (const S*)res; a.out_scale = scale;
        return launch<DT, EPI_RES>(a, (hipStream_t)stream, "primx_linear_residual");
    });
    return PRIMX_OK;
}

// (rowops.hip)
int primx_launch_ln_modulate(const float* x, const void* shift, const void* scale, int64_t mod_stride, void* out, int dtype,
                             int rows, int rows_per_batch, int D, float eps, const void* pf0, int64_t pf0_bytes,
                             const void* pf1, int64_t pf1_bytes, void* stream, const char* name);

extern "C" int primx_linear_gate_residual_ln(const void* A, const void* W, const void* bias, const void* gate,
                                             int64_t gate_stride, float* x, int M, int N, int K, int rows_per_batch,
                                             const void* ln_shift, const void* ln_scale, int64_t ln_mod_stride, void* ln_out,
                                             float ln_eps, void* sync, int64_t sync_words, int dtype, const void* prefetch,
                                             int64_t prefetch_bytes, void* stream) {
    const char* name = ln_out ? "primx_linear_gate_residual_ln" : "primx_linear_gate_residual";
    PRIMX_REQUIRE(gate && x && rows_per_batch > 0, "%s: bad argument", name);
    PRIMX_REQUIRE(!ln_out || (ln_shift && ln_scale), "%s: ln_out needs ln_shift and ln_scale", name);
    PRIMX_REQUIRE(!sync || sync_words >= 2 * (int64_t)((M + 127) / 128), "%s: sync needs two words per 128-row block", name);
    g_ln_fused = false;
    PRIMX_DISPATCH_16(dtype, name, {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W; a.bias = (const S*)bias;
        a.M = M; a.N = N; a.K = K;
        a.gate = (const S*)gate; a.gate_stride = gate_stride; a.x = x; a.rows_per_batch = rows_per_batch;
        a.ln_shift = (const S*)ln_shift; a.ln_scale = (const S*)ln_scale; a.ln_mod_stride = ln_mod_stride;
        a.ln_out = (S*)ln_out; a.ln_eps = ln_eps; a.sync = (unsigned*)sync;
        if (int rc = set_prefetch<DT>(a, prefetch, prefetch_bytes, name)) return rc;
        if (int rc = launch<DT, EPI_GATE_RESIDUAL>(a, (hipStream_t)stream, name)) return rc;
    });
    if (ln_out && !g_ln_fused)   // shapes the tail does not cover: the same LayerNorm as a launch of its own
        return primx_launch_ln_modulate(x, ln_shift, ln_scale, ln_mod_stride, ln_out, dtype, M, rows_per_batch, N, ln_eps, nullptr, 0,
                                        nullptr, 0, stream, name);
    return PRIMX_OK;
}

extern "C" int primx_linear_gate_residual(const void* A, const void* W, const void* bias, const void* gate,
                                          int64_t gate_stride, float* x, int M, int N, int K, int rows_per_batch,
                                          int dtype, const void* prefetch, int64_t prefetch_bytes, void* stream) {
    return primx_linear_gate_residual_ln(A, W, bias, gate, gate_stride, x, M, N, K, rows_per_batch, nullptr, nullptr, 0, nullptr,
                                         0.f, nullptr, 0, dtype, prefetch, prefetch_bytes, stream);
}

extern "C" int primx_linear_heads(const void* A, const void* W, const void* bias, int M, int N, int K,
                                  int rows_per_batch, int heads, int dh, int n_seg, const int* kind, void* const* dst,
                                  int n_rep, int rep_batches, int n_pad, float scale0, int dtype, const void* prefetch,
                                  int64_t prefetch_bytes, void* stream) {
    PRIMX_REQUIRE(kind && dst && n_seg >= 1 && n_seg <= 3 && n_rep >= 1, "primx_linear_heads: n_seg must be 1..3, n_rep >= 1");
    PRIMX_REQUIRE(heads > 0 && dh > 0 && N == n_rep * n_seg * heads * dh,
                  "primx_linear_heads: N must equal n_rep*n_seg*heads*dh");
    PRIMX_REQUIRE(rows_per_batch > 0 && M % rows_per_batch == 0 && n_pad >= rows_per_batch && n_pad % 16 == 0,
                  "primx_linear_heads: need M %% rows_per_batch == 0, n_pad >= rows_per_batch, n_pad %% 16 == 0");
    for (int s = 0; s < n_seg; ++s) {
        PRIMX_REQUIRE(dst[s] != nullptr, "primx_linear_heads: null destination");
        PRIMX_REQUIRE(kind[s] == PRIMX_HEADS_ROWS || kind[s] == PRIMX_HEADS_VT || kind[s] == PRIMX_HEADS_KROWS,
                      "primx_linear_heads: bad kind");
    }
    PRIMX_DISPATCH_16(dtype, "primx_linear_heads", {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W; a.bias = (const S*)bias;
        a.M = M; a.N = N; a.K = K;
        a.rows_per_batch = rows_per_batch; a.heads = heads; a.dh = dh; a.DP = primx_padded_head_dim(dh);
        a.n_pad = n_pad; a.n_seg = n_seg; a.scale0 = scale0;
        for (int s = 0; s < 3; ++s) {
            a.kind[s] = s < n_seg ? kind[s] : 0;
            // one (batch, head) block holds n_pad * row-stride elements in the token-major layouts, DP * n_pad in VT
            a.rep_stride[s] = (int64_t)rep_batches * heads * n_pad * (a.kind[s] == PRIMX_HEADS_VT ? a.DP : heads_row_stride(a.kind[s], a.DP));
            a.dst[s] = s < n_seg ? (S*)dst[s] : nullptr;
        }
        if (int rc = set_prefetch<DT>(a, prefetch, prefetch_bytes, "primx_linear_heads")) return rc;
        return launch<DT, EPI_HEADS>(a, (hipStream_t)stream, "primx_linear_heads");
    });
    return PRIMX_OK;
}

// ---- Grouped few-row fp32-out Linear (ABI 26): the fold's u / v rows of EVERY site of a planned loop from one launch.  Each problem is
// out_i[M, N_i] = A_i[M, K] W_i[N_i, K]^T (+ bias_i for the rows >= bias_from_row) with the same M (2 x timesteps of the loop: 50 rows at
// 25 steps) and K: 83 problems at DiT-XL, 592 MB of weights read once - a weight stream, where 83 separate launches of 8 - 32 workgroups
// were latency-bound (~18 us each: 1.5 ms per 25-step loop).  A workgroup = 4 waves x 32 columns of ONE problem and 64 rows (grid.y walks
// the row blocks of a longer loop); a wave streams its 32 weight rows straight into the MFMA's first operand (accumulator = C^T: a lane
// owns one A row and four consecutive columns, 16-byte fp32 stores) and reads the A fragments through L1 / L2 (64 x K halves per problem,
// shared by its workgroups).  k runs in order in one fp32 accumulator per output: same products, same order for every (M, problem count).
template <int DT>
__global__ __launch_bounds__(256) void f32out_group_kernel(const PrimxF32outProblem* __restrict__ probs, int n_probs, int M, int K,
                                                            int bias_from_row) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the problem of this workgroup: the last one whose first workgroup is <= blockIdx.x (uniform binary search over <= a few hundred entries)
    int lo = 0, hi = n_probs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (probs[mid].first_wg <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PrimxF32outProblem pr = probs[lo];
    const int n0 = ((int)blockIdx.x - pr.first_wg) * 128 + wave * 32;
    if (n0 >= pr.N) return;                                      // (N % 32 == 0: a wave's 32 columns are all inside or all outside)
    const int m0 = blockIdx.y * 64;
    const S* A = static_cast<const S*>(pr.A);
    const S* W = static_cast<const S*>(pr.W);
    const S* arow[4];
    const S* wrow[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) arow[i] = A + (int64_t)min(m0 + i * 16 + lr, M - 1) * K + lg * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) wrow[j] = W + (int64_t)(n0 + j * 16 + lr) * K + lg * 8;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int UN = 4;                                        // k-steps of 32 in flight: 4 x (2 + 4) 16-byte loads per lane
    int k = 0;
    for (; k + 32 * UN <= K; k += 32 * UN) {
        V8 a[UN][4], b[UN][2];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int j = 0; j < 2; ++j) b[u][j] = __builtin_nontemporal_load(reinterpret_cast<const V8*>(wrow[j] + k + 32 * u));
#pragma unroll
            for (int i = 0; i < 4; ++i) a[u][i] = *reinterpret_cast<const V8*>(arow[i] + k + 32 * u);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = T16<DT>::mfma16(b[u][j], a[u][i], acc[i][j]);
    }
    for (; k < K; k += 32) {
        V8 a[4], b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = __builtin_nontemporal_load(reinterpret_cast<const V8*>(wrow[j] + k));
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const V8*>(arow[i] + k);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = T16<DT>::mfma16(b[j], a[i], acc[i][j]);
    }
    // acc[i][j][r] = C[m0 + 16 i + lr][n0 + 16 j + 4 lg + r]
    const S* bias = static_cast<const S*>(pr.bias);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 16 + 4 * lg;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = (float)bias[n + r];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + i * 16 + lr;
            if (m >= M) continue;
            f32x4 o = acc[i][j];
            if (m >= bias_from_row) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += bv[r];
            }
            *reinterpret_cast<f32x4*>(pr.out + (int64_t)m * pr.N + n) = o;
        }
    }
}

// `probs`: n_probs descriptors in DEVICE memory, first_wg ascending from 0 with first_wg[i + 1] - first_wg[i] = ceil(N_i / 128); total_wg =
// their sum.  (The descriptors are data the kernel reads: the host cannot validate them here - the caller builds them, cf. ops.F32outGroup.)
extern "C" int primx_linear_f32out_group(const PrimxF32outProblem* probs, int n_probs, int total_wg, int M, int K, int bias_from_row,
                                         int dtype, void* stream) {
    const char* name = "primx_linear_f32out_group";
    PRIMX_REQUIRE(probs && n_probs > 0 && total_wg >= n_probs, "%s: null or empty problem list", name);
    PRIMX_REQUIRE(M > 0 && K > 0 && K % 32 == 0 && bias_from_row >= 0, "%s: need M > 0, K %% 32 == 0 (M=%d K=%d)", name, M, K);
    PRIMX_DISPATCH_16(dtype, name, {
        PRIMX_NOTE_KERNEL("f32out_group_kernel<%d>", DT);
        hipLaunchKernelGGL((f32out_group_kernel<DT>), dim3(total_wg, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, probs, n_probs, M, K,
                           bias_from_row);
    });
    PRIMX_CHECK_LAUNCH(name);
    return PRIMX_OK;
}

// ---- LayerNorm fold (see fold_stats_load)
extern "C" int primx_linear_f32out(const void* A, const void* W, const void* bias, float* out, int M, int N, int K,
                                   int bias_from_row, int dtype, void* stream) {
    PRIMX_REQUIRE(A && W && out, "primx_linear_f32out: null pointer");
    PRIMX_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0, "primx_linear_f32out: need M,N>0 and K %% 8 == 0 (M=%d N=%d K=%d)", M, N, K);
    PRIMX_DISPATCH_16(dtype, "primx_linear_f32out", {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W; a.bias = (const S*)bias;
        a.M = M; a.N = N; a.K = K;
        a.x = out; a.rows_per_batch = bias_from_row;
        // the loader-wave 128 x 144 kernel where the shape allows (N % 144 == 0, K % 64 == 0; PRIMX_F32OUT_TILE144=0: never) - the fold's
        // u / v rows are 8 - 32 workgroups of 18 k-tiles, latency-bound on the register-staged generic kernel (~20 us each); same box,
        // the configs[1] step: 8.453 - 8.471 -> 8.446 - 8.452 ms (tools/gpu/r4_uv144.sh)
        static const bool tile144 = [] { const char* e = getenv("PRIMX_F32OUT_TILE144"); return !(e && e[0] == '0'); }();
        if (tile144 && g_loader && N % 144 == 0 && K % BK == 0 && (((uintptr_t)out | (uintptr_t)bias) & 15) == 0) {
            PRIMX_NOTE_KERNEL("gemm144l_dma_kernel<%d, %d>", DT, EPI_F32OUT);
            hipLaunchKernelGGL((gemm144l_dma_kernel<DT, EPI_F32OUT>), dim3(((M + 127) / 128) * (N / 144)), dim3(640), 0,
                               (hipStream_t)stream, PRIMX_GEMM_PASS(a));
        } else {
            PRIMX_NOTE_KERNEL("gemm_kernel<%d, %d, 32, 2, 2, 2, 2, 0>", DT, EPI_F32OUT);
            hipLaunchKernelGGL((gemm_kernel<DT, EPI_F32OUT, 32, 2, 2, 2, 2, 0>), dim3(((M + 127) / 128) * ((N + 127) / 128)), dim3(256), 0,
                               (hipStream_t)stream, a);
        }
    });
    PRIMX_CHECK_LAUNCH("primx_linear_f32out");
    return PRIMX_OK;
}

extern "C" int primx_linear_gate_residual_fold(const void* A, const void* W, const void* bias, const void* gate,
                                               int64_t gate_stride, float* x, int M, int N, int K, int rows_per_batch,
                                               const void* next_scale, int64_t next_mod_stride, const float* center,
                                               void* a16_out, float* part_out, int dtype, const void* prefetch,
                                               int64_t prefetch_bytes, void* stream) {
    const char* name = "primx_linear_gate_residual_fold";
    PRIMX_REQUIRE(gate && x && rows_per_batch > 0 && next_scale && center && a16_out && part_out, "%s: bad argument", name);
    PRIMX_DISPATCH_16(dtype, name, {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W; a.bias = (const S*)bias;
        a.M = M; a.N = N; a.K = K;
        a.gate = (const S*)gate; a.gate_stride = gate_stride; a.x = x; a.rows_per_batch = rows_per_batch;
        a.ln_scale = (const S*)next_scale; a.ln_mod_stride = next_mod_stride; a.ln_out = (S*)a16_out;
        a.fold_c = center; a.fold_part = part_out;
        if (int rc = set_prefetch<DT>(a, prefetch, prefetch_bytes, name)) return rc;
        return launch_fold<DT, EPI_GATE_RESIDUAL_FOLD>(a, (hipStream_t)stream, name);
    });
    return PRIMX_OK;
}

extern "C" int primx_linear_heads_fold(const void* A, const void* W, int M, int N, int K, int rows_per_batch, int heads, int dh,
                                       int n_seg, const int* kind, void* const* dst, int n_pad, float scale0, const float* part,
                                       const float* u, const float* v, const float* center, float* center_out, float eps,
                                       int dtype, const void* prefetch, int64_t prefetch_bytes, void* stream) {
    const char* name = "primx_linear_heads_fold";
    PRIMX_REQUIRE(kind && dst && n_seg >= 1 && n_seg <= 3, "%s: n_seg must be 1..3", name);
    PRIMX_REQUIRE(heads > 0 && dh > 0 && N == n_seg * heads * dh, "%s: N must equal n_seg*heads*dh", name);
    PRIMX_REQUIRE(rows_per_batch > 0 && M % rows_per_batch == 0 && n_pad >= rows_per_batch && n_pad % 16 == 0,
                  "%s: need M %% rows_per_batch == 0, n_pad >= rows_per_batch, n_pad %% 16 == 0", name);
    PRIMX_REQUIRE(part && u && v && center && center_out && center != center_out, "%s: null fold argument, or center_out == center", name);
    for (int s = 0; s < n_seg; ++s) {
        PRIMX_REQUIRE(dst[s] != nullptr, "%s: null destination", name);
        PRIMX_REQUIRE(kind[s] == PRIMX_HEADS_ROWS || kind[s] == PRIMX_HEADS_VT || kind[s] == PRIMX_HEADS_KROWS, "%s: bad kind", name);
    }
    PRIMX_DISPATCH_16(dtype, name, {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W;
        a.M = M; a.N = N; a.K = K;
        a.rows_per_batch = rows_per_batch; a.heads = heads; a.dh = dh; a.DP = primx_padded_head_dim(dh);
        a.n_pad = n_pad; a.n_seg = n_seg; a.scale0 = scale0;
        for (int s = 0; s < 3; ++s) {
            a.kind[s] = s < n_seg ? kind[s] : 0;
            a.rep_stride[s] = 0;
            a.dst[s] = s < n_seg ? (S*)dst[s] : nullptr;
        }
        a.fold_part = const_cast<float*>(part); a.fold_parts = K / 144; a.fold_u = u; a.fold_v = v; a.fold_c = center; a.fold_c_out = center_out; a.fold_eps = eps;
        if (int rc = set_prefetch<DT>(a, prefetch, prefetch_bytes, name)) return rc;
        return launch_fold<DT, EPI_HEADS_FOLD>(a, (hipStream_t)stream, name);
    });
    return PRIMX_OK;
}

// Does a heads problem map onto the 256 x 288 tile's LDS-staged scatter epilogue (the shape rule of launch<> / launch_fold, without their
// workgroup-count thresholds)?
template <int DT>
static bool big_heads_shape(const GemmArgs<DT>& a) {
    const int per = a.heads * a.dh;
    return !g_no_big && g_big_heads_min > 0 && a.heads > 0 && a.N % 288 == 0 && per % 288 == 0 && 288 % a.dh == 0 && a.dh % 8 == 0 &&
           a.dh >= 32 && a.rows_per_batch % 256 == 0 && a.M % 256 == 0 && a.K % BK == 0;
}

// primx_linear_heads_fold (problem 0, no carried prefetch) and primx_linear_heads (problem 1: n_rep = 1, no carried prefetch) from ONE
// launch: problem 1's tiles run on the CUs problem 0's single round of 256 x 288 tiles leaves idle (gemm288q_pair_kernel).  A == NULL:
// problem 1 alone, on the same tile kernel - so that its bits never depend on whether it rode.  Where the pairing rule does not hold
// the two launches are made one after the other: same results.
extern "C" int primx_linear_heads_fold_pair(const void* A, const void* W, int M, int N, int K, int rows_per_batch, int heads, int dh,
                                            int n_seg, const int* kind, void* const* dst, int n_pad, float scale0, const float* part,
                                            const float* u, const float* v, const float* center, float* center_out, float eps,
                                            const void* A2, const void* W2, const void* bias2, int M2, int N2, int K2,
                                            int rows_per_batch2, int heads2, int dh2, int n_seg2, const int* kind2, void* const* dst2,
                                            int n_pad2, float scale0_2, int dtype, void* stream) {
    const char* name = "primx_linear_heads_fold_pair";
    PRIMX_REQUIRE(A2 && W2 && kind2 && dst2 && n_seg2 >= 1 && n_seg2 <= 3, "%s: problem 1: null operand, or n_seg outside 1..3", name);
    PRIMX_REQUIRE(heads2 > 0 && dh2 > 0 && N2 == n_seg2 * heads2 * dh2 && M2 > 0 && K2 > 0, "%s: problem 1: N must equal n_seg*heads*dh", name);
    PRIMX_REQUIRE(rows_per_batch2 > 0 && M2 % rows_per_batch2 == 0 && n_pad2 >= rows_per_batch2 && n_pad2 % 16 == 0,
                  "%s: problem 1: need M %% rows_per_batch == 0, n_pad >= rows_per_batch, n_pad %% 16 == 0", name);
    for (int s = 0; s < n_seg2; ++s)
        PRIMX_REQUIRE(dst2[s] != nullptr && (kind2[s] == PRIMX_HEADS_ROWS || kind2[s] == PRIMX_HEADS_VT || kind2[s] == PRIMX_HEADS_KROWS),
                      "%s: problem 1: null destination or bad kind", name);
    bool paired = false;
    if (A) {
        PRIMX_REQUIRE(W && kind && dst && n_seg >= 1 && n_seg <= 3 && heads > 0 && dh > 0 && N == n_seg * heads * dh && M > 0 && K > 0 &&
                          rows_per_batch > 0 && M % rows_per_batch == 0 && n_pad >= rows_per_batch && n_pad % 16 == 0 && part && u && v &&
                          center && center_out && center != center_out,
                      "%s: problem 0: the argument rules of primx_linear_heads_fold", name);
        for (int s = 0; s < n_seg; ++s) PRIMX_REQUIRE(dst[s] != nullptr, "%s: problem 0: null destination", name);
    }
    PRIMX_DISPATCH_16(dtype, name, {
        using S = typename T16<DT>::S;
        GemmArgs<DT> r = {};
        r.A = (const S*)A2; r.W = (const S*)W2; r.bias = (const S*)bias2;
        r.M = M2; r.N = N2; r.K = K2;
        r.rows_per_batch = rows_per_batch2; r.heads = heads2; r.dh = dh2; r.DP = primx_padded_head_dim(dh2);
        r.n_pad = n_pad2; r.n_seg = n_seg2; r.scale0 = scale0_2;
        for (int s = 0; s < 3; ++s) {
            r.kind[s] = s < n_seg2 ? kind2[s] : 0;
            r.rep_stride[s] = 0;
            r.dst[s] = s < n_seg2 ? (S*)dst2[s] : nullptr;
        }
        const bool big1 = big_heads_shape<DT>(r);
        const int g1 = big1 ? (r.M / 256) * (r.N / 288) : 0;
        if (big1) r.xcd_gm = big_xcd_gm<DT>(r, true);
        if (A) {
            GemmArgs<DT> a = {};
            a.A = (const S*)A; a.W = (const S*)W;
            a.M = M; a.N = N; a.K = K;
            a.rows_per_batch = rows_per_batch; a.heads = heads; a.dh = dh; a.DP = primx_padded_head_dim(dh);
            a.n_pad = n_pad; a.n_seg = n_seg; a.scale0 = scale0;
            for (int s = 0; s < 3; ++s) {
                a.kind[s] = s < n_seg ? kind[s] : 0;
                a.rep_stride[s] = 0;
                a.dst[s] = s < n_seg ? (S*)dst[s] : nullptr;
            }
            a.fold_part = const_cast<float*>(part); a.fold_parts = K / 144; a.fold_u = u; a.fold_v = v; a.fold_c = center; a.fold_c_out = center_out;
            a.fold_eps = eps;
            const int g0 = big_heads_shape<DT>(a) ? (a.M / 256) * (a.N / 288) : 0;
            // one round: both problems' tiles at once on the 256 CUs; problem 0 at the size launch_fold gives the big tile; the fold rules
            if (big1 && g0 >= g_big_heads_min && g0 % 8 == 0 && g0 + g1 <= 256 && !g_gemm_prof_on && K % 144 == 0 && a.fold_parts <= 8 &&
                (((uintptr_t)u | (uintptr_t)v) & 15) == 0 && (((uintptr_t)part | (uintptr_t)center | (uintptr_t)center_out) & 7) == 0) {
                a.xcd_gm = big_xcd_gm<DT>(a, true);
                a.ln_light = g_ln_mode;
                if (heads_kt64<DT>(a) && heads_kt64<DT>(r)) {
                    PRIMX_NOTE_KERNEL("gemm288q_pair_kernel<%d, 64>", DT);
                    hipLaunchKernelGGL((gemm288q_pair_kernel<DT, 64>), dim3(g0 + g1), dim3(512), 0, (hipStream_t)stream, PRIMX_GEMM_PASS(a), g0, r);
                } else {
                    PRIMX_NOTE_KERNEL("gemm288q_pair_kernel<%d, 32>", DT);
                    hipLaunchKernelGGL((gemm288q_pair_kernel<DT, 32>), dim3(g0 + g1), dim3(512), 0, (hipStream_t)stream, PRIMX_GEMM_PASS(a), g0, r);
                }
                PRIMX_CHECK_LAUNCH(name);
                paired = true;
            } else if (int rc = launch_fold<DT, EPI_HEADS_FOLD>(a, (hipStream_t)stream, name)) {
                return rc;
            }
        }
        if (!paired) {
            if (big1) {   // problem 1 alone: the tile kernel it would have ridden
                launch288q<DT, EPI_HEADS>(r, dim3(g1), (hipStream_t)stream);
                PRIMX_CHECK_LAUNCH(name);
            } else {
                return launch<DT, EPI_HEADS>(r, (hipStream_t)stream, name);
            }
        }
    });
    return PRIMX_OK;
}

extern "C" int primx_linear_fold(const void* A, const void* W, void* out, int M, int N, int K, int act, const float* part,
                                 const float* u, const float* v, const float* center, float* center_out, float eps, int dtype,
                                 const void* prefetch, int64_t prefetch_bytes, void* stream) {
    const char* name = "primx_linear_fold";
    PRIMX_REQUIRE(out && part && u && v && center && center_out && center != center_out, "%s: null pointer, or center_out == center", name);
    PRIMX_REQUIRE(act == PRIMX_ACT_NONE || act == PRIMX_ACT_GELU_TANH || act == PRIMX_ACT_GELU_ERF, "%s: bad activation code", name);
    PRIMX_DISPATCH_16(dtype, name, {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W;
        a.M = M; a.N = N; a.K = K;
        a.out = (S*)out; a.act = act; a.out_scale = 1.0f;
        a.fold_part = const_cast<float*>(part); a.fold_parts = K / 144; a.fold_u = u; a.fold_v = v; a.fold_c = center; a.fold_c_out = center_out; a.fold_eps = eps;
        if (int rc = set_prefetch<DT>(a, prefetch, prefetch_bytes, name)) return rc;
        return launch_fold<DT, EPI_LINEAR_FOLD>(a, (hipStream_t)stream, name);
    });
    return PRIMX_OK;
}

extern "C" int primx_conv3d_k3(const void* in, const void* Wk, const void* bias, const void* res, float res_scale,
                               void* out, int P, int S, int Cin, int Cout, int Kpad, int dtype, void* stream) {
    PRIMX_REQUIRE(in && Wk && out, "primx_conv3d_k3: null pointer");
    const int cl = ilog2_exact(Cin);
    PRIMX_REQUIRE(P > 0 && S > 0 && Cout > 0 && cl >= 3, "primx_conv3d_k3: Cin must be a power of two >= 8 (Cin=%d)", Cin);
    PRIMX_REQUIRE(Kpad >= 27 * Cin && Kpad % BK == 0, "primx_conv3d_k3: Kpad must be a multiple of 64 >= 27*Cin");
    PRIMX_DISPATCH_16(dtype, "primx_conv3d_k3", {
        using Sx = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const Sx*)in; a.W = (const Sx*)Wk; a.bias = (const Sx*)bias;
        a.M = P * S * S * S; a.N = Cout; a.K = Kpad;
        a.out = (Sx*)out; a.res = (const Sx*)res; a.out_scale = res_scale;
        a.S3 = S; a.cin_log2 = cl;
        return launch<DT, EPI_RES, 1>(a, (hipStream_t)stream, "primx_conv3d_k3");
    });
    return PRIMX_OK;
}

extern "C" int primx_convtranspose_k2s2(const void* in, const void* Wt, const void* bias, void* out, int P, int S,
                                        int Cin, int Cout, int dtype, void* stream) {
    PRIMX_REQUIRE(in && Wt && out, "primx_convtranspose_k2s2: null pointer");
    PRIMX_REQUIRE(P > 0 && S > 0 && Cout > 0 && Cin % 8 == 0, "primx_convtranspose_k2s2: Cin %% 8 != 0");
    PRIMX_DISPATCH_16(dtype, "primx_convtranspose_k2s2", {
        using Sx = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const Sx*)in; a.W = (const Sx*)Wt; a.bias = (const Sx*)bias;
        a.M = P * S * S * S; a.N = 8 * Cout; a.K = Cin;
        a.out = (Sx*)out; a.S3 = S; a.cout = Cout;
        return launch<DT, EPI_CONVT>(a, (hipStream_t)stream, "primx_convtranspose_k2s2");
    });
    return PRIMX_OK;
}
