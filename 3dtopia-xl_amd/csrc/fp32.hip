// The reference's fp32 call path (DiT.forward(x, t, y) with the signature defaults precision_dtype=float32,
// enable_amp=False - models/dit_crossattn.py:184, and `precision: tf32` in inference.py:239-247) on gfx950.
//
// gfx950 has no TF32 / xf32 matrix instruction, so this path is EXACT fp32: every contraction runs on
// v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, bitwise an fmaf chain; 157 TFLOP/s peak = 1/16 of the fp16 rate),
// softmax / LayerNorm / GELU in fp32 VALU.  That is at least the precision of the reference's own fp32 run (true fp32 in
// app.py, TF32 tensor cores under the CLI's allow_tf32 flags).  Tolerance vs the fp32 golden: rel-L2 <= 1e-4 (tests).
//
// Kernels: (1) gemm_f32_kernel     out = act(A W^T + b) * s           or   out += gate[b] * (A W^T + bias)   (fp32)
//          (2) attn_f32_kernel     softmax(q k^T * scale) v on strided [B, M, H, dh] views (xformers BMHK semantics)
//          (3) ln_modulate_f32     LN(x) * (1 + scale) + shift
//          (4) silu_f32
// The 64-cycle fp32 MFMA hides all LDS / global traffic of these simple single-buffered loops; nothing here is tuned
// further - the 16-bit autocast path (gemm.hip / attention.hip) is the production path.
#include "common.h"

// D[i][j] += sum_{k<2} A[i][k] B[k][j]: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// accumulator register r of lane l is D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
static __device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// tanh-GELU in fp32 as torch evaluates it (no fast-math shortcuts: this path is the exact one)
static __device__ __forceinline__ float gelu_tanh_exact(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}

// ---------------------------------------------------------------------------------------------------------------------
// (1) fp32 GEMM, 128 x 128 x 16 tiles, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles.
// LDS images are k-major ([k][row], row stride 132 floats): the fragment read of step s is one ds_read_b32 per lane at
// consecutive addresses within each half-wave (conflict-free).  One tile of register prefetch.
constexpr int GF_BM = 128, GF_BN = 128, GF_BK = 16, GF_LD = GF_BM + 4;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* out, int M, int N, int K,
                                                       int act, float out_scale, const float* __restrict__ gate,
                                                       int64_t gate_stride, int rows_per_batch) {
    __shared__ float As[GF_BK][GF_LD];
    __shared__ float Bs[GF_BK][GF_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GF_BM, n0 = blockIdx.x * GF_BN;
    const int li = lane & 31, hi = lane >> 5;
    // loader: 128 rows x 4 float4 per operand tile = 512 float4, two per thread
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    f32x4 ra[2], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = lrow + 64 * p;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[p] = (m0 + row < M && k0 + lk < K) ? *reinterpret_cast<const f32x4*>(A + (int64_t)(m0 + row) * K + k0 + lk) : z;
            rb[p] = (n0 + row < N && k0 + lk < K) ? *reinterpret_cast<const f32x4*>(W + (int64_t)(n0 + row) * K + k0 + lk) : z;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = lrow + 64 * p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                As[lk + j][row] = ra[p][j];
                Bs[lk + j][row] = rb[p][j];
            }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += GF_BK) {
        __syncthreads();  // previous tile's fragment reads are done
        stash();
        __syncthreads();
        if (k0 + GF_BK < K) fetch(k0 + GF_BK);
#pragma unroll
        for (int s = 0; s < GF_BK / 2; ++s) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[2 * s + hi][wm * 64 + i * 32 + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[2 * s + hi][wn * 64 + j * 32 + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f32(a[i], b[j], acc[i][j]);
        }
    }
    // epilogue: lane = one output column, registers walk the rows -> 128-byte coalesced row segments
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + li;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + acc_row(r, hi);
                if (m >= M) continue;
                float v = acc[i][j][r] + bv;
                float* o = out + (int64_t)m * N + n;
                if (gate) {
                    const float g = gate[(int64_t)(m / rows_per_batch) * gate_stride + n];
                    *o = *o + g * v;       // x = x + gate * branch   (dit_crossattn.py:55-57), fp32 throughout
                } else {
                    if (act == PRIMX_ACT_GELU_TANH) v = gelu_tanh_exact(v);
                    else if (act == PRIMX_ACT_GELU_ERF) v = gelu_erf_f(v);
                    *o = v * out_scale;
                }
            }
        }
    }
}

extern "C" int primx_gemm_f32(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int act,
                              float out_scale, const float* gate, int64_t gate_stride, int rows_per_batch, void* stream) {
    PRIMX_REQUIRE(A && W && out, "primx_gemm_f32: null pointer");
    PRIMX_REQUIRE(M > 0 && N > 0 && K > 0 && K % 4 == 0, "primx_gemm_f32: need K %% 4 == 0 (K=%d)", K);
    PRIMX_REQUIRE(act >= 0 && act <= 2, "primx_gemm_f32: bad activation %d", act);
    PRIMX_REQUIRE(!gate || rows_per_batch > 0, "primx_gemm_f32: gate needs rows_per_batch > 0");
    PRIMX_REQUIRE(((uintptr_t)A | (uintptr_t)W) % 16 == 0, "primx_gemm_f32: A and W must be 16-byte aligned");
    dim3 grid((N + GF_BN - 1) / GF_BN, (M + GF_BM - 1) / GF_BM);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, W, bias, out, M, N, K, act, out_scale,
                       gate, gate_stride, rows_per_batch > 0 ? rows_per_batch : 1);
    PRIMX_CHECK_LAUNCH("primx_gemm_f32");
    return PRIMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// (2) fp32 attention.  Workgroup = 4 waves x 32 queries; key tiles of 32 staged in LDS and shared by the waves.
//   S^T[key][q]  = K Q^T   (A = K rows from LDS, B = the wave's Q in registers) -> lane = one query, 16 keys per lane:
//                  row max / row sum are per-lane scalars + ONE lane^32 exchange.
//   O^T[d][q]   += V^T P^T (A = V^T from LDS, B = P straight from the S registers: contraction step s takes key
//                  acc_row(s, hi) on half `hi`, which is exactly the key register s of that half holds).
// NT = number of 32-row tiles of the head dim (dh <= 32 NT).
template <int NT>
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ out, int H, int Nq,
                                                       int Nkv, int dh, int64_t qb, int64_t qm, int64_t qh, int64_t kb,
                                                       int64_t km, int64_t kh, int64_t vb, int64_t vm, int64_t vh,
                                                       float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int DPAD = 32 * NT;          // padded head dim of the V image and of the O accumulators
    constexpr int KS = DPAD + 1;           // K image row stride (odd: the 32 keys of a fragment read hit 32 banks)
    float* Ks = smem;                      // [32][KS]
    float* Vs = smem + 32 * KS;            // [32][DPAD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qrow = blockIdx.x * 128 + wave * 32 + li;
    const int nsteps = (dh + 1) / 2;
    const float* qp = q + b * qb + h * qh + (int64_t)(qrow < Nq ? qrow : Nq - 1) * qm;
    float qreg[16 * NT];
#pragma unroll
    for (int s = 0; s < 16 * NT; ++s) {
        const int d = 2 * s + hi;
        qreg[s] = d < dh ? qp[d] : 0.f;
    }
    f32x16 O[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
    float m_run = -INFINITY, l_lane = 0.f;
    const float* kbase = k + b * kb + h * kh;
    const float* vbase = v + b * vb + h * vh;
    const int lrow = tid >> 5, lcol = tid & 31;   // loader: 8 rows per pass, 32 consecutive floats per row segment

    for (int key0 = 0; key0 < Nkv; key0 += 32) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = lrow + 8 * p, key = key0 + row;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const int d = lcol + 32 * c;
                const bool ok = key < Nkv && d < dh;
                Ks[row * KS + d] = ok ? kbase[(int64_t)key * km + d] : 0.f;
                Vs[row * DPAD + d] = ok ? vbase[(int64_t)key * vm + d] : 0.f;
            }
        }
        __syncthreads();
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16 * NT; ++s)
            if (s < nsteps) S = mfma_f32(Ks[li * KS + 2 * s + hi], qreg[s], S);
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S[r] = key0 + acc_row(r, hi) < Nkv ? S[r] * scale : -INFINITY;
            mt = fmaxf(mt, S[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);        // finite: every tile holds at least one valid key
        const float alpha = __expf(m_run - m_new);   // 0 on the first tile (m_run = -inf)
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S[r] = __expf(S[r] - m_new);
            psum += S[r];
        }
        l_lane = l_lane * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float* vr = Vs + acc_row(s, hi) * DPAD + li;
#pragma unroll
            for (int t = 0; t < NT; ++t) O[t] = mfma_f32(vr[32 * t], S[s], O[t]);
        }
    }
    const float inv = 1.0f / (l_lane + __shfl_xor(l_lane, 32));
    if (qrow < Nq) {
        float* op = out + ((int64_t)(b * Nq + qrow) * H + h) * dh;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * t + acc_row(r, hi);
                if (d < dh) op[d] = O[t][r] * inv;
            }
    }
}

extern "C" int primx_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int Nq,
                                   int Nkv, int dh, const int64_t* q_strides, const int64_t* k_strides,
                                   const int64_t* v_strides, float scale, void* stream) {
    PRIMX_REQUIRE(q && k && v && out && q_strides && k_strides && v_strides, "primx_attention_f32: null pointer");
    PRIMX_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nkv > 0 && dh > 0 && dh <= 128, "primx_attention_f32: bad shape (dh <= 128)");
    PRIMX_REQUIRE(B <= 65535 && H <= 65535, "primx_attention_f32: B, H must fit a grid dimension");
    const int NT = (dh + 31) / 32;
    dim3 grid((Nq + 127) / 128, H, B);
    const size_t lds = (size_t)32 * (32 * NT + 1 + 32 * NT) * sizeof(float);
#define AF_LAUNCH(NT_)                                                                                                \
    hipLaunchKernelGGL((attn_f32_kernel<NT_>), grid, dim3(256), lds, (hipStream_t)stream, q, k, v, out, H, Nq, Nkv, dh, \
                       q_strides[0], q_strides[1], q_strides[2], k_strides[0], k_strides[1], k_strides[2], v_strides[0], \
                       v_strides[1], v_strides[2], scale)
    switch (NT) {
        case 1: AF_LAUNCH(1); break;
        case 2: AF_LAUNCH(2); break;
        case 3: AF_LAUNCH(3); break;
        default: AF_LAUNCH(4); break;
    }
#undef AF_LAUNCH
    PRIMX_CHECK_LAUNCH("primx_attention_f32");
    return PRIMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// (3) LayerNorm (no affine) + modulate, fp32 in and out: one wave per row, two-pass statistics over registers.
__global__ __launch_bounds__(256) void ln_modulate_f32_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                              const float* __restrict__ scale, int64_t mod_stride,
                                                              float* __restrict__ out, int rows, int rows_per_batch, int D,
                                                              float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * D;
    constexpr int MAXC = 32;               // D <= 2048
    float vals[MAXC];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int d = lane + 64 * c;
        vals[c] = d < D ? xr[d] : 0.f;
        sum += vals[c];
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int d = lane + 64 * c;
        const float dv = d < D ? vals[c] - mean : 0.f;
        sq += dv * dv;
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    const int64_t mb = (int64_t)(row / rows_per_batch) * mod_stride;
    float* orow = out + (int64_t)row * D;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int d = lane + 64 * c;
        if (d < D) orow[d] = (vals[c] - mean) * rstd * (1.0f + scale[mb + d]) + shift[mb + d];
    }
}

extern "C" int primx_layernorm_modulate_f32(const float* x, const float* shift, const float* scale, int64_t mod_stride,
                                            float* out, int rows, int rows_per_batch, int D, float eps, void* stream) {
    PRIMX_REQUIRE(x && shift && scale && out, "primx_layernorm_modulate_f32: null pointer");
    PRIMX_REQUIRE(rows > 0 && rows_per_batch > 0 && D > 0 && D <= 2048, "primx_layernorm_modulate_f32: need 0 < D <= 2048");
    hipLaunchKernelGGL(ln_modulate_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, shift, scale,
                       mod_stride, out, rows, rows_per_batch, D, eps);
    PRIMX_CHECK_LAUNCH("primx_layernorm_modulate_f32");
    return PRIMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// (4) SiLU, fp32 (the nn.SiLU in front of every adaLN Linear, dit_crossattn.py:40-43,69-72)
__global__ void silu_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = in[i];
        out[i] = v / (1.0f + expf(-v));
    }
}

extern "C" int primx_silu_f32(const float* in, float* out, int64_t n, void* stream) {
    PRIMX_REQUIRE(in && out && n > 0, "primx_silu_f32: bad argument");
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(silu_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, n);
    PRIMX_CHECK_LAUNCH("primx_silu_f32");
    return PRIMX_OK;
}
