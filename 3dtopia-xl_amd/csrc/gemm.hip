// 16-bit MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (both operands K-contiguous: activations row-major,
// weights in nn.Linear (out,in) layout) with the DiT block's and the VAE decoder's epilogues fused.
//
// 256 threads = 4 waves of v_mfma_f32_32x32x16_{f16,bf16}; two tile shapes:
//   Wide  : 128 x 128 x 64, waves 2x2, each wave 64x64 (2x2 MFMA tiles)   - every DiT Linear, 256-ch convs
//   Narrow: 128 x  32 x 64, waves 4x1, each wave 32x32 (1 MFMA tile)      - Cout <= 32 convs of the VAE
// Operand orientation: MFMA-A = activation rows (m), MFMA-B = weight rows (n), so the accumulator's
// lane index (lane & 31) runs along n - the contiguous dimension of every destination - and its
// registers run along m.  LDS rows are padded 64 -> 72 halves (144 B): 144/16 = 9 is odd, so the 16
// lanes of a ds_read_b128 group (distinct rows mod 16) hit 16 distinct 16-byte slots - conflict-free
// (MI355X_MICROARCH.md, LDS).  Register-staged double buffering: the global loads of k-tile t+1 are in
// flight while the MFMAs of k-tile t run; one barrier per k-tile.  Workgroup ids are remapped so that
// each XCD (private 4 MiB L2) owns a contiguous run of tiles that share A row-panels.
//
// GATHER = 1 turns the A loader into the implicit-GEMM gather of a 3x3x3 / stride 1 / pad 1
// convolution over channels-last [P, S^3, Cin] activations: k = tap * Cin + ci, row m = (p, voxel);
// out-of-volume taps (and the zero-padded tail of K) read a 16-byte zero block instead of branching.
#include "common.h"

namespace {

// 16 zero bytes: the source of every out-of-range operand chunk (K tail, conv padding taps)
__device__ const u32x4 g_zero16 = {0u, 0u, 0u, 0u};

constexpr int BK = 64;
constexpr int LROW = BK + 8;  // padded LDS row (halves)

enum { EPI_LINEAR = 0, EPI_GATE_RESIDUAL = 1, EPI_HEADS = 2, EPI_RES = 3, EPI_CONVT = 4 };

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
    S* dst[3];
    float scale0;
    // EPI_RES: out = cast16((acc + bias + res) * out_scale); res may be null
    const S* res;
    // conv gather (GATHER) and EPI_CONVT geometry
    int S3;               // grid edge S (volume S^3)
    int cin_log2;         // log2(Cin)
    int cout;             // EPI_CONVT: N = 8 * cout
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "contiguous chunk per XCD" remap (cdna_hip_programming.md T1)
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// WM x WN waves, each MI x NI tiles of 32x32
template <int DT, int EPI, int WM, int WN, int MI, int NI, int GATHER>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs<DT> p) {
    using S = typename T16<DT>::S;
    using V8 = typename T16<DT>::V8;
    using V4 = typename T16<DT>::V4;
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr int TA = BM * LROW, TW = BN * LROW;   // halves per operand tile
    constexpr int NA = BM * 8 / 256, NW = (BN * 8 + 255) / 256;  // 16-byte chunks per thread
    __shared__ __attribute__((aligned(16))) S smem[2 * (TA + TW)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    const int nt = (p.N + BN - 1) / BN, mt = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, nt * mt);
    const int m0 = (id / nt) * BM, n0 = (id % nt) * BN;

    // ---- loader geometry: chunk c = tid + 256*i -> row c>>3, 16-byte column c&7
    const int kc = (tid & 7) * 8;
    const S* ga[NA];
    const S* gw[NW];
    int offa[NA], offw[NW];
    int gz[NA], gy[NA], gx[NA];       // GATHER: voxel coordinates of the row
    const S* gbase[NA];               // GATHER: &in[p, 0, 0]
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (tid + 256 * i) >> 3;
        const int ra = min(m0 + row, p.M - 1);  // clamp: rows >= M are computed but never stored
        offa[i] = row * LROW + kc;
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
        const int row = (tid + 256 * i) >> 3;
        const int rw = min(n0 + row, p.N - 1);
        gw[i] = p.W + (int64_t)rw * p.K + kc;
        offw[i] = row * LROW + kc;
    }
    const bool w_active = (NW * 256 <= BN * 8) || (tid < BN * 8);  // Narrow: one chunk per thread, all active

    const S* zeros = reinterpret_cast<const S*>(&g_zero16);
    auto load_a = [&](int kt, V8 (&r)[NA]) {
        if (GATHER) {
            const int kk = kt * BK + kc;
            const int tap = kk >> p.cin_log2, ci = kk & ((1 << p.cin_log2) - 1);
            const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int z = gz[i] + dz, y = gy[i] + dy, x = gx[i] + dx;
                const bool ok = tap < 27 && (unsigned)z < (unsigned)p.S3 && (unsigned)y < (unsigned)p.S3 &&
                                (unsigned)x < (unsigned)p.S3;
                const S* src = gbase[i] + ((int64_t)((z * p.S3 + y) * p.S3 + x) << p.cin_log2) + ci;
                r[i] = *reinterpret_cast<const V8*>(ok ? src : zeros);
            }
        } else {
            const bool k_ok = kt * BK + kc < p.K;  // K tail (K % 64 != 0): zero chunk
#pragma unroll
            for (int i = 0; i < NA; ++i) r[i] = *reinterpret_cast<const V8*>(k_ok ? ga[i] + kt * BK : zeros);
        }
    };
    auto load_w = [&](int kt, V8 (&r)[NW]) {
        const bool k_ok = kt * BK + kc < p.K;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (w_active) r[i] = *reinterpret_cast<const V8*>(k_ok ? gw[i] + kt * BK : zeros);
    };
    auto store_tiles = [&](int buf, V8 (&ra)[NA], V8 (&rw)[NW]) {
        S* base = smem + buf * (TA + TW);
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<V8*>(base + offa[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (w_active) *reinterpret_cast<V8*>(base + TA + offw[i]) = rw[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    V8 ra_[NA], rw_[NW];
    load_a(0, ra_);
    load_w(0, rw_);
    store_tiles(0, ra_, rw_);
    __syncthreads();

    const int nk = (p.K + BK - 1) / BK;
    const int a_rd = (wm * MI * 32 + l31) * LROW + hi * 8;
    const int w_rd = (wn * NI * 32 + l31) * LROW + hi * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            load_a(kt + 1, ra_);
            load_w(kt + 1, rw_);
        }
        const S* As = smem + (kt & 1) * (TA + TW);
        const S* Ws = As + TA;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            V8 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const V8*>(As + a_rd + i * 32 * LROW + s * 16);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const V8*>(Ws + w_rd + j * 32 * LROW + s * 16);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = T16<DT>::mfma32(a[i], b[j], acc[i][j]);
        }
        if (more) store_tiles((kt + 1) & 1, ra_, rw_);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    // acc[mi][ni][r] = C[m0 + (wm*MI + mi)*32 + (r&3) + 8*(r>>2) + 4*hi][n0 + (wn*NI + ni)*32 + l31]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + (wn * NI + ni) * 32 + l31;
        const bool n_ok = n < p.N;
        int seg = 0, hh = 0, dd = 0;  // EPI_HEADS: column -> (segment, head, d);  EPI_CONVT: seg = tap, dd = co
        if (EPI == EPI_HEADS && n_ok) {
            const int per = p.heads * p.dh;
            seg = n / per;
            const int w = n - seg * per;
            hh = w / p.dh;
            dd = w - hh * p.dh;
        }
        if (EPI == EPI_CONVT && n_ok) {
            seg = n / p.cout;
            dd = n - seg * p.cout;
        }
        const float bv = (p.bias && n_ok) ? (float)p.bias[EPI == EPI_CONVT ? dd : n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int mbase = m0 + (wm * MI + mi) * 32 + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // register quad g: rows mbase + 8g + {0,1,2,3}
                const int mq = mbase + 8 * g;
                if (!n_ok) continue;
                if (EPI == EPI_LINEAR) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = mq + j;
                        if (m >= p.M) continue;
                        float y = rnd16<DT>(acc[mi][ni][4 * g + j] + bv);
                        if (p.act == PRIMX_ACT_GELU_TANH) y = rnd16<DT>(gelu_tanh_f(y));
                        if (p.out_scale != 1.0f) y = rnd16<DT>(p.out_scale * y);
                        p.out[(int64_t)m * p.N + n] = (S)y;
                    }
                } else if (EPI == EPI_RES) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = mq + j;
                        if (m >= p.M) continue;
                        float y = acc[mi][ni][4 * g + j] + bv;
                        if (p.res) y += (float)p.res[(int64_t)m * p.N + n];
                        p.out[(int64_t)m * p.N + n] = (S)(y * p.out_scale);
                    }
                } else if (EPI == EPI_CONVT) {
                    // row m = (prim, z, y, x) on the S^3 grid; column = (tap dz,dy,dx ; co) -> voxel (2z+dz, ..)
                    const int Sg = p.S3, V = Sg * Sg * Sg, S2 = 2 * Sg;
                    const int dz = seg >> 2, dy = (seg >> 1) & 1, dx = seg & 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = mq + j;
                        if (m >= p.M) continue;
                        const int pp = m / V, v = m - pp * V;
                        const int z = v / (Sg * Sg), y = (v / Sg) % Sg, x = v % Sg;
                        const int64_t ov = ((int64_t)(2 * z + dz) * S2 + (2 * y + dy)) * S2 + (2 * x + dx);
                        p.out[((int64_t)pp * 8 * V + ov) * p.cout + dd] = (S)(acc[mi][ni][4 * g + j] + bv);
                    }
                } else if (EPI == EPI_GATE_RESIDUAL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = mq + j;
                        if (m >= p.M) continue;
                        const int b = m / p.rows_per_batch;
                        const float gt = (float)p.gate[(int64_t)b * p.gate_stride + n];
                        const float v = rnd16<DT>(acc[mi][ni][4 * g + j] + bv);
                        float* xp = p.x + (int64_t)m * p.N + n;
                        *xp = *xp + rnd16<DT>(gt * v);
                    }
                } else {  // EPI_HEADS
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = rnd16<DT>(acc[mi][ni][4 * g + j] + bv);
                    S* dst = p.dst[0];
                    int kind = p.kind[0];
                    if (seg == 1) { dst = p.dst[1]; kind = p.kind[1]; }
                    if (seg == 2) { dst = p.dst[2]; kind = p.kind[2]; }
                    if (seg == 0 && p.scale0 != 1.0f) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = rnd16<DT>(p.scale0 * v[j]);
                    }
                    const bool quad_ok = (p.rows_per_batch % 4 == 0) && (mq + 3 < p.M);
                    if (kind == PRIMX_HEADS_VT && quad_ok) {
                        // 4 consecutive tokens of one batch entry = one contiguous quad of the VT layout
                        const int b = mq / p.rows_per_batch, tok = mq - b * p.rows_per_batch;
                        const int64_t head = (int64_t)b * p.heads + hh;
                        V4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (S)v[j];
                        *reinterpret_cast<V4*>(dst + (head * p.DP + dd) * p.n_pad + vt_key_pos(tok)) = o;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int m = mq + j;
                            if (m >= p.M) continue;
                            const int b = m / p.rows_per_batch, tok = m - b * p.rows_per_batch;
                            const int64_t head = (int64_t)b * p.heads + hh;
                            if (kind == PRIMX_HEADS_ROWS) dst[(head * p.n_pad + tok) * p.DP + dd] = (S)v[j];
                            else dst[(head * p.DP + dd) * p.n_pad + vt_key_pos(tok)] = (S)v[j];
                        }
                    }
                }
            }
        }
    }
}

template <int DT, int EPI, int GATHER = 0>
int launch(const GemmArgs<DT>& a, hipStream_t st, const char* name) {
    PRIMX_REQUIRE(a.A && a.W, "%s: null operand", name);
    PRIMX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 8 == 0, "%s: need M,N>0 and K %% 8 == 0 (M=%d N=%d K=%d)",
                  name, a.M, a.N, a.K);
    const int mt = (a.M + 127) / 128;
    if (a.N <= 32) {  // Narrow tile
        hipLaunchKernelGGL((gemm_kernel<DT, EPI, 4, 1, 1, 1, GATHER>), dim3(mt * ((a.N + 31) / 32)), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((gemm_kernel<DT, EPI, 2, 2, 2, 2, GATHER>), dim3(mt * ((a.N + 127) / 128)), dim3(256), 0, st,
                           a);
    }
    PRIMX_CHECK_LAUNCH(name);
    return PRIMX_OK;
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace

extern "C" int primx_linear(const void* A, const void* W, const void* bias, void* out, int M, int N, int K, int dtype,
                            int act, float out_scale, void* stream) {
    PRIMX_REQUIRE(out, "primx_linear: null output");
    PRIMX_REQUIRE(act == PRIMX_ACT_NONE || act == PRIMX_ACT_GELU_TANH, "primx_linear: bad activation code");
    PRIMX_DISPATCH_16(dtype, "primx_linear", {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W; a.bias = (const S*)bias;
        a.M = M; a.N = N; a.K = K;
        a.out = (S*)out; a.act = act; a.out_scale = out_scale;
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

extern "C" int primx_linear_gate_residual(const void* A, const void* W, const void* bias, const void* gate,
                                          int64_t gate_stride, float* x, int M, int N, int K, int rows_per_batch,
                                          int dtype, void* stream) {
    PRIMX_REQUIRE(gate && x && rows_per_batch > 0, "primx_linear_gate_residual: bad argument");
    PRIMX_DISPATCH_16(dtype, "primx_linear_gate_residual", {
        using S = typename T16<DT>::S;
        GemmArgs<DT> a = {};
        a.A = (const S*)A; a.W = (const S*)W; a.bias = (const S*)bias;
        a.M = M; a.N = N; a.K = K;
        a.gate = (const S*)gate; a.gate_stride = gate_stride; a.x = x; a.rows_per_batch = rows_per_batch;
        return launch<DT, EPI_GATE_RESIDUAL>(a, (hipStream_t)stream, "primx_linear_gate_residual");
    });
    return PRIMX_OK;
}

extern "C" int primx_linear_heads(const void* A, const void* W, const void* bias, int M, int N, int K,
                                  int rows_per_batch, int heads, int dh, int n_seg, const int* kind, void* const* dst,
                                  int n_pad, float scale0, int dtype, void* stream) {
    PRIMX_REQUIRE(kind && dst && n_seg >= 1 && n_seg <= 3, "primx_linear_heads: n_seg must be 1..3");
    PRIMX_REQUIRE(heads > 0 && dh > 0 && N == n_seg * heads * dh, "primx_linear_heads: N must equal n_seg*heads*dh");
    PRIMX_REQUIRE(rows_per_batch > 0 && M % rows_per_batch == 0 && n_pad >= rows_per_batch && n_pad % 16 == 0,
                  "primx_linear_heads: need M %% rows_per_batch == 0, n_pad >= rows_per_batch, n_pad %% 16 == 0");
    for (int s = 0; s < n_seg; ++s) {
        PRIMX_REQUIRE(dst[s] != nullptr, "primx_linear_heads: null destination");
        PRIMX_REQUIRE(kind[s] == PRIMX_HEADS_ROWS || kind[s] == PRIMX_HEADS_VT, "primx_linear_heads: bad kind");
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
            a.dst[s] = s < n_seg ? (S*)dst[s] : nullptr;
        }
        return launch<DT, EPI_HEADS>(a, (hipStream_t)stream, "primx_linear_heads");
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
