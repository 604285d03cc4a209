// PrimSDF field query: the O(points x primitives) evaluation behind mesh / texture extraction
// (models/primsdf.py:52-109, driven by inference.py:106-116 over 256^3 points and 180-193 over the visible texels).
//
//   w_i(x)   = relu(1 - ||(x - pos_i) / scale_i||_inf)                        (prim_weight, primsdf.py:103-107)
//   out(x)   = sum_i [w_i / (sum_j w_j + 1e-6)] * trilinear(feat_i, (x - pos_i) / scale_i)   (grid_sample_feat, 66-76;
//              grid_sample 'bilinear', align_corners=True: x -> W (fastest), y -> H, z -> D of the [6, S, S, S] volume)
//   eval mode, points no primitive covers: nearest primitive by ||x - pos||_2, nearest of its S^3 grid points,
//              sdf = s + dist * sign(s) with s the stored SDF there; the other channels stay 0               (78-100)
//   preds: sdf = out[0], tex = clip(out[1:4], 0, 1), mat = clip(out[4:6], 0, 1)                          (forward, 52-64)
//
// One thread per point, primitives (scale, pos) streamed through LDS in chunks; the feature volumes (P x 6 x S^3 fp32,
// 25 MB at P = 2048, S = 8) stay in L2 / MALL and only the handful of covering primitives of a point are sampled.
// max_d fl(|x_d - pos_d| / s) == fl(max_d |x_d - pos_d| / s) (rounding is monotonic), so the cull needs no division and
// the weight one division per covering primitive - identical values to the reference's per-axis divisions.
#include "common.h"

namespace {

constexpr int CHUNK = 1024;   // primitives per LDS chunk (16 KB)

__global__ __launch_bounds__(256) void primsdf_query_kernel(const float* __restrict__ pts, const float* __restrict__ srt,
                                                           const float* __restrict__ feat, const float* __restrict__ lin,
                                                           float* __restrict__ out, int n, int P, int S, int C,
                                                           int eval_fill) {
    __shared__ float4 prim[CHUNK];   // (scale, x, y, z)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) { x = pts[3 * (int64_t)i]; y = pts[3 * (int64_t)i + 1]; z = pts[3 * (int64_t)i + 2]; }
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f, best = 3.4e38f;
    int best_i = 0;
    const int S3 = S * S * S;
    const float half = 0.5f * (float)(S - 1);
    for (int p0 = 0; p0 < P; p0 += CHUNK) {
        const int cnt = min(CHUNK, P - p0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) prim[t] = reinterpret_cast<const float4*>(srt)[p0 + t];
        __syncthreads();
        if (!live) continue;
        for (int t = 0; t < cnt; ++t) {
            const float4 q = prim[t];
            const float dx = x - q.y, dy = y - q.z, dz = z - q.w;
            if (eval_fill) {
                const float d2 = dx * dx + dy * dy + dz * dz;   // argmin of the L2 distance (first minimum wins)
                if (d2 < best) { best = d2; best_i = p0 + t; }
            }
            const float m = fmaxf(fabsf(dx), fmaxf(fabsf(dy), fabsf(dz)));
            if (!(m < q.x)) continue;                          // w = relu(1 - m / s) = 0
            const float w = 1.0f - m / q.x;
            if (!(w > 0.f)) continue;
            wsum += w;
            // grid_sample, align_corners=True: index = (coord + 1) / 2 * (S - 1)
            const float fx = (dx / q.x + 1.0f) * half, fy = (dy / q.x + 1.0f) * half, fz = (dz / q.x + 1.0f) * half;
            int ix = min((int)floorf(fx), S - 2), iy = min((int)floorf(fy), S - 2), iz = min((int)floorf(fz), S - 2);
            ix = max(ix, 0); iy = max(iy, 0); iz = max(iz, 0);
            const float tx = fx - (float)ix, ty = fy - (float)iy, tz = fz - (float)iz;
            const float* vol = feat + (int64_t)(p0 + t) * C * S3 + (iz * S + iy) * S + ix;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if (c >= C) break;
                const float* v = vol + c * S3;
                const float c00 = v[0] * (1.f - tx) + v[1] * tx;
                const float c01 = v[S] * (1.f - tx) + v[S + 1] * tx;
                const float c10 = v[S * S] * (1.f - tx) + v[S * S + 1] * tx;
                const float c11 = v[S * S + S] * (1.f - tx) + v[S * S + S + 1] * tx;
                const float s = (c00 * (1.f - ty) + c01 * ty) * (1.f - tz) + (c10 * (1.f - ty) + c11 * ty) * tz;
                acc[c] += w * s;
            }
        }
    }
    if (!live) return;
    float r[6];
    const float inv = 1.0f / (wsum + 1e-6f);
#pragma unroll
    for (int c = 0; c < 6; ++c) r[c] = acc[c] * inv;
    if (eval_fill && !(wsum > 0.f)) {
        const float4 q = reinterpret_cast<const float4*>(srt)[best_i];
        // nearest grid point of the nearest primitive: the distance is separable, so per axis
        const float px[3] = {x, y, z}, pc[3] = {q.y, q.z, q.w};
        int idx[3];
        float d2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int bi = 0;
            float bd = 3.4e38f;
            for (int k = 0; k < S; ++k) {
                const float e = px[a] - (pc[a] + q.x * lin[k]);
                if (e * e < bd) { bd = e * e; bi = k; }
            }
            idx[a] = bi;
            d2 += bd;
        }
        const float s = feat[(int64_t)best_i * C * S3 + (idx[2] * S + idx[1]) * S + idx[0]];   // channel 0, [z][y][x]
        const float sgn = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
        r[0] = s + sqrtf(d2) * sgn;
    }
    float* o = out + (int64_t)i * C;
    o[0] = r[0];
#pragma unroll
    for (int c = 1; c < 6; ++c)
        if (c < C) o[c] = fminf(fmaxf(r[c], 0.f), 1.f);
}

}  // namespace

extern "C" int primx_primsdf_query(const float* pts, const float* srt, const float* feat, const float* lin, float* out,
                                   int n, int P, int S, int C, int eval_fill, void* stream) {
    PRIMX_REQUIRE(pts && srt && feat && lin && out, "primx_primsdf_query: null pointer");
    PRIMX_REQUIRE(n > 0 && P > 0 && S >= 2 && S <= 32 && C >= 1 && C <= 6, "primx_primsdf_query: need n, P > 0, 2 <= S <= 32, 1 <= C <= 6");
    hipLaunchKernelGGL(primsdf_query_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts, srt, feat,
                       lin, out, n, P, S, C, eval_fill);
    PRIMX_CHECK_LAUNCH("primx_primsdf_query");
    return PRIMX_OK;
}
