// Volumetric primitive ray marcher, forward only: what RayMarcher.forward computes for the
// previews / turntable of the CLI and the app (dva/ray_marcher.py:142-229 -> compute_raydirs + mvpraymarch with
// algo = 0, chlast template, no warp field, SRT primitive transforms, additive accumulation, "fixedorder" BVH).
//
// Semantics followed (reference CUDA, dva/mvp/extensions/):
//   rays          utils/utils_kernel.cu:15-56     origin = campos / volradius, dir = normalize(R^T-rows * ((pix - princpt) / focal, 1)),
//                                                 [tmin, tmax] = slab test against [-1, 1]^3, tmin clamped to 0
//   hit list      mvpraymarch/utils.h:728-824     fixed-order tree = primitives in INDEX order; a primitive enters the list of
//                                                 every ray of the warp if any ray's local slab test passes (at most 512);
//                                                 each ray keeps the union [rtmin, rtmax] of ITS OWN intersections
//   marching      mvpraymarch_subset_kernel.h:58-86   t starts at tmin + floor((rtmin - tmin) / dt) * dt; per step, per listed
//                                                 primitive in order: y = (R (x - pos)) * scale; if |y| < 1 strictly, not
//                                                 saturated and t < rtmax + 1e-5: sample, accumulate
//   sample        primsampler.h:37-62 + utils.h:406-500   trilinear, align_corners, zero padding, channels-last float4 voxels;
//                                                 alpha *= exp(-fadescale * sum |y_d|^fadeexp)
//   accumulation  primaccum.h:63-78               newalpha = a + alpha * dt; rgba += (rgb, 1) * (min(newalpha, 1) - a); saturate at 1
//
// MI355X design: one wave = one 8 x 8 pixel tile.  There is no tree: the 64 rays of a tile test ALL K primitives (the
// primitive record is wave-uniform, so it comes through scalar loads) and build the tile's hit list in LDS with a ballot
// - 2048 primitives x 270 k rays is ~1 ms of VALU, less than the tree build + traversal it replaces; the list order is the
// index order the reference's "fixedorder" tree produces, so saturation happens at the same sample.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

// PRIMX_RAYMARCH_STATS=1: per-wave counters [waves, sum num, marching waves, chunks, sum nsub, steps, record evaluations]
__device__ unsigned long long g_rm_stats[12];

constexpr int MAXHIT = 512;   // the reference's maxhitboxes
constexpr int CHUNK = 96;     // marching steps per sub-list rebuild
constexpr int NREC = 64;      // primitives per wave and chunk with an LDS record (16 floats) and per-ray entry / exit steps

struct F3 { float x, y, z; };
__device__ __forceinline__ float fast_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ F3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

__global__ __launch_bounds__(256) void raydirs_kernel(const float* __restrict__ viewpos, const float* __restrict__ viewrot,
                                                     const float* __restrict__ focal, const float* __restrict__ princpt,
                                                     const float* __restrict__ pixelcoords, float volradius,
                                                     float* __restrict__ raypos, float* __restrict__ raydir,
                                                     float* __restrict__ tminmax, int N, int H, int W) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * H * W) return;
    const int w = (int)(idx % W), h = (int)((idx / W) % H), n = (int)(idx / ((int64_t)W * H));
    const F3 rp = {viewpos[n * 3] / volradius, viewpos[n * 3 + 1] / volradius, viewpos[n * 3 + 2] / volradius};
    float px = (float)w, py = (float)h;
    if (pixelcoords) { px = pixelcoords[idx * 2]; py = pixelcoords[idx * 2 + 1]; }
    px = (px - princpt[n * 2]) / focal[n * 2];
    py = (py - princpt[n * 2 + 1]) / focal[n * 2 + 1];
    const float* R = viewrot + n * 9;
    F3 d = {R[0] * px + R[3] * py + R[6], R[1] * px + R[4] * py + R[7], R[2] * px + R[5] * py + R[8]};
    const float inv = rsqrtf(d.x * d.x + d.y * d.y + d.z * d.z);     // helper_math normalize(): v * rsqrtf(dot(v, v))
    d = {d.x * inv, d.y * inv, d.z * inv};
    const float t1x = (-1.f - rp.x) / d.x, t2x = (1.f - rp.x) / d.x;
    const float t1y = (-1.f - rp.y) / d.y, t2y = (1.f - rp.y) / d.y;
    const float t1z = (-1.f - rp.z) / d.z, t2z = (1.f - rp.z) / d.z;
    const float tmin = fmaxf(fminf(t1x, t2x), fmaxf(fminf(t1y, t2y), fminf(t1z, t2z)));
    const float tmax = fminf(fmaxf(t1x, t2x), fminf(fmaxf(t1y, t2y), fmaxf(t1z, t2z)));
    raypos[idx * 3] = rp.x; raypos[idx * 3 + 1] = rp.y; raypos[idx * 3 + 2] = rp.z;
    raydir[idx * 3] = d.x; raydir[idx * 3 + 1] = d.y; raydir[idx * 3 + 2] = d.z;
    tminmax[idx * 2] = fmaxf(tmin, 0.f);
    tminmax[idx * 2 + 1] = tmax;
}

// block = 4 waves = 16 x 16 pixels, each wave an 8 x 8 tile with its own hit list
__global__ __launch_bounds__(256) void raymarch_kernel(const float* __restrict__ rayposim, const float* __restrict__ raydirim,
                                                      const float* __restrict__ tminmaxim, float stepsize,
                                                      const float* __restrict__ primpos, const float* __restrict__ primrot,
                                                      const float* __restrict__ primscale, const float4* __restrict__ tplate,
                                                      float4* __restrict__ rayrgba, int N, int H, int W, int K, int TD, int TH,
                                                      int TW, float fadescale, float fadeexp, int stats) {
    __shared__ int hits[4][MAXHIT];
    __shared__ int subs[4][MAXHIT];
    __shared__ float4 recs[4][NREC][4];   // (pos.xyz, r00) (r01 r02 r10 r11) (r12 r20 r21 r22) (scale.xyz, -)
    __shared__ unsigned short span[4][NREC][64];   // per ray: first | last << 8 step of the chunk the ray can be inside
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.z;
    const int w = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int h = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool valid = w < W && h < H;
    const int64_t pix = ((int64_t)n * H + min(h, H - 1)) * W + min(w, W - 1);   // out-of-image lanes shadow an edge ray
    F3 rp = ld3(rayposim + pix * 3);
    const F3 rd = ld3(raydirim + pix * 3);
    const float tmin0 = tminmaxim[pix * 2], tmax0 = tminmaxim[pix * 2 + 1];
    const float* ppos = primpos + (int64_t)n * K * 3;
    const float* prot = primrot + (int64_t)n * K * 9;
    const float* pscl = primscale + (int64_t)n * K * 3;
    const float4* tpl = tplate + (int64_t)n * K * TD * TH * TW;

    const unsigned long long c_start = stats ? __builtin_readcyclecounter() : 0;
    // ---- hit list of the tile (index order) + this ray's own [rtmin, rtmax]
    int* list = hits[wave];
    int num = 0;
    float rtmin = INFINITY, rtmax = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const float* pr = prot + k * 9;
        const F3 xm = {rp.x - ppos[k * 3], rp.y - ppos[k * 3 + 1], rp.z - ppos[k * 3 + 2]};
        // PrimTransfSRT::forward2: r = (pr0 x + pr1 y + pr2 z) * scale   (pr_i = i-th row of the stored 3 x 3)
        const float sx = pscl[k * 3], sy = pscl[k * 3 + 1], sz = pscl[k * 3 + 2];
        const F3 r0 = {(pr[0] * xm.x + pr[3] * xm.y + pr[6] * xm.z) * sx, (pr[1] * xm.x + pr[4] * xm.y + pr[7] * xm.z) * sy,
                       (pr[2] * xm.x + pr[5] * xm.y + pr[8] * xm.z) * sz};
        const F3 r1 = {(pr[0] * rd.x + pr[3] * rd.y + pr[6] * rd.z) * sx, (pr[1] * rd.x + pr[4] * rd.y + pr[7] * rd.z) * sy,
                       (pr[2] * rd.x + pr[5] * rd.y + pr[8] * rd.z) * sz};
        const float ix = 1.0f / r1.x, iy = 1.0f / r1.y, iz = 1.0f / r1.z;
        const float ax = (-1.f - r0.x) * ix, bx = (1.f - r0.x) * ix;
        const float ay = (-1.f - r0.y) * iy, by = (1.f - r0.y) * iy;
        const float az = (-1.f - r0.z) * iz, bz = (1.f - r0.z) * iz;
        const float trmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float trmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        const bool hit = trmin <= trmax;
        if (hit) { rtmin = fminf(rtmin, trmin); rtmax = fmaxf(rtmax, trmax); }
        if (__any(hit) && num < MAXHIT) {       // wave-uniform
            if (lane == 0) list[num] = k;
            ++num;
        }
    }
    rtmin = fmaxf(rtmin, tmin0);
    rtmax = fminf(rtmax, tmax0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // the list (written by lane 0) is read by the whole wave below
    __builtin_amdgcn_wave_barrier();

    const unsigned long long c_hit = stats ? __builtin_readcyclecounter() : 0;
    unsigned long long c_rebuild = 0, c_t0 = 0;
    // ---- march
    const F3 ro = rp;                       // ray origin (t = 0) for the per-chunk interval tests
    int* sub = subs[wave];
    float t = tmin0;
    rp = {rp.x + rd.x * tmin0, rp.y + rd.y * tmin0, rp.z + rd.z * tmin0};
    const int incs = (int)floorf((rtmin - t) / stepsize);
    t += incs * stepsize;
    rp = {rp.x + rd.x * incs * stepsize, rp.y + rd.y * incs * stepsize, rp.z + rd.z * incs * stepsize};
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    bool sat = false;
    const int sD = TW * TH, sH = TW;
    // The shipped step is 1e-4 of the volume: a ray takes ~10^4 steps while a primitive spans a few hundred of them.
    // Every CHUNK steps the tile's hit list is filtered down to the primitives whose (per-ray) slab interval overlaps the
    // chunk for ANY ray of the wave, with a two-step safety margin; inside the chunk only those are transformed and
    // tested.  The filter is conservative, and a listed primitive the sample is not inside fails `in` exactly as before,
    // so the image is the one the unfiltered loop produces.
    unsigned long long st_chunks = 0, st_nsub = 0, st_steps = 0, st_eval = 0;
    while (!__all(t > rtmax + 1e-5f || sat)) {
        if (stats) c_t0 = __builtin_readcyclecounter();
        const float t_end = t + (float)CHUNK * stepsize;
        const bool live = !(t > rtmax + 1e-5f || sat);
        int nsub = 0;
        for (int ks = 0; ks < num; ++ks) {
            const int k = __builtin_amdgcn_readfirstlane(list[ks]);   // wave-uniform: the primitive record comes through scalar loads
            const float* pr = prot + k * 9;
            const F3 xm = {ro.x - ppos[k * 3], ro.y - ppos[k * 3 + 1], ro.z - ppos[k * 3 + 2]};
            const float sx = pscl[k * 3], sy = pscl[k * 3 + 1], sz = pscl[k * 3 + 2];
            const F3 r0 = {(pr[0] * xm.x + pr[3] * xm.y + pr[6] * xm.z) * sx, (pr[1] * xm.x + pr[4] * xm.y + pr[7] * xm.z) * sy,
                           (pr[2] * xm.x + pr[5] * xm.y + pr[8] * xm.z) * sz};
            const F3 r1 = {(pr[0] * rd.x + pr[3] * rd.y + pr[6] * rd.z) * sx, (pr[1] * rd.x + pr[4] * rd.y + pr[7] * rd.z) * sy,
                           (pr[2] * rd.x + pr[5] * rd.y + pr[8] * rd.z) * sz};
            const float ix = 1.0f / r1.x, iy = 1.0f / r1.y, iz = 1.0f / r1.z;
            const float ax = (-1.f - r0.x) * ix, bx = (1.f - r0.x) * ix;
            const float ay = (-1.f - r0.y) * iy, by = (1.f - r0.y) * iy;
            const float az = (-1.f - r0.z) * iz, bz = (1.f - r0.z) * iz;
            const float trmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
            const float trmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
            // NaNs (ray parallel to a slab it touches) compare false everywhere: keep such a primitive
            const bool skip = (trmax < t - 2.f * stepsize) || (trmin > t_end + 2.f * stepsize) || (trmin > trmax + 4.f * stepsize);
            if (__any(live && !skip)) {
                if (lane == 0) sub[nsub] = k;
                if (nsub < NREC) {
                    // conservative step window of THIS ray inside primitive k during the chunk (two steps of margin each side;
                    // anything that does not compare cleanly - NaN, infinities - becomes "always")
                    float e0 = floorf((trmin - t) / stepsize) - 2.f, e1 = ceilf((trmax - t) / stepsize) + 2.f;
                    unsigned short w = 0x00ff;                                   // first 255 > last 0: never
                    if (live && !skip) {
                        const int i0 = (e0 >= 0.f && e0 <= 255.f) ? (int)e0 : (e0 > 255.f ? 255 : 0);
                        const int i1 = (e1 >= 0.f && e1 <= 255.f) ? (int)e1 : (e1 < 0.f ? 0 : 255);
                        w = (unsigned short)(i0 | (i1 << 8));
                        if (!(trmin <= trmax)) w = 0xff00;                       // unordered: evaluate every step
                    }
                    span[wave][nsub][lane] = w;
                }
                ++nsub;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        // The records of the chunk's cached primitives go to LDS once.  Per step a ray only compares the step index with
        // its window (2 bytes per ray per primitive, lane-linear LDS read); the record is fetched (LDS broadcast) and the
        // exact inside test / sample run only when some ray of the wave is within its window.
        st_chunks += 1; st_nsub += nsub;
        if (stats) c_rebuild += __builtin_readcyclecounter() - c_t0;
        const int ncache = min(nsub, NREC);
        float* recf = reinterpret_cast<float*>(recs[wave]);
        for (int i = lane; i < ncache * 16; i += 64) {
            const int k = sub[i >> 4], j = i & 15;
            float v = 0.f;
            if (j < 3) v = ppos[k * 3 + j];
            else if (j < 12) v = prot[k * 9 + j - 3];
            else if (j < 15) v = pscl[k * 3 + j - 12];
            recf[i] = v;
        }
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_wave_barrier();
        for (int step = 0; step < CHUNK; ++step) {
            if (__all(t > rtmax + 1e-5f || sat)) break;
            st_steps += 1;
            for (int ks = 0; ks < nsub; ++ks) {
                float4 c0, c1, c2, c3;
                if (ks < NREC) {
                    const unsigned w = span[wave][ks][lane];
                    const bool maybe = step >= (int)(w & 255u) && step <= (int)(w >> 8);
                    if (!__any(maybe)) continue;          // nobody near this primitive at this step: 6 instructions
                    c0 = recs[wave][ks][0]; c1 = recs[wave][ks][1]; c2 = recs[wave][ks][2]; c3 = recs[wave][ks][3];
                } else {   // beyond the cache (rare): straight from memory, every step
                    const int k = __builtin_amdgcn_readfirstlane(sub[ks]);
                    const float* pr_ = prot + k * 9;
                    c0 = {ppos[k * 3], ppos[k * 3 + 1], ppos[k * 3 + 2], pr_[0]};
                    c1 = {pr_[1], pr_[2], pr_[3], pr_[4]};
                    c2 = {pr_[5], pr_[6], pr_[7], pr_[8]};
                    c3 = {pscl[k * 3], pscl[k * 3 + 1], pscl[k * 3 + 2], 0.f};
                }
                const int k = __builtin_amdgcn_readfirstlane(sub[ks]);
                st_eval += 1;
                const float pr[9] = {c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w};
                const F3 xm = {rp.x - c0.x, rp.y - c0.y, rp.z - c0.z};
                const float yx = (pr[0] * xm.x + pr[3] * xm.y + pr[6] * xm.z) * c3.x;
                const float yy = (pr[1] * xm.x + pr[4] * xm.y + pr[7] * xm.z) * c3.y;
                const float yz = (pr[2] * xm.x + pr[5] * xm.y + pr[8] * xm.z) * c3.z;
                const bool in = yx > -1.f && yx < 1.f && yy > -1.f && yy < 1.f && yz > -1.f && yz < 1.f;
                if (in && !sat && t < rtmax + 1e-5f) {
                    // the reference's __powf / __expf are CUDA's fast forms 2^(y log2 x) and 2^(x log2 e); HIP's __powf is
                    // the full-precision routine (~200 instructions - measured: 1900 VALU instructions per marching step)
                    const float fade = fast_exp(-fadescale * (fast_pow(fabsf(yx), fadeexp) + fast_pow(fabsf(yy), fadeexp) +
                                                              fast_pow(fabsf(yz), fadeexp)));
                    const float gx = (yx + 1.f) * 0.5f * (float)(TW - 1), gy = (yy + 1.f) * 0.5f * (float)(TH - 1),
                                gz = (yz + 1.f) * 0.5f * (float)(TD - 1);
                    const int x0 = min(max((int)floorf(gx), 0), TW - 2), y0 = min(max((int)floorf(gy), 0), TH - 2),
                              z0 = min(max((int)floorf(gz), 0), TD - 2);   // (no-op clamps: memory safety only)
                    const float fx = gx - (float)x0, fy = gy - (float)y0, fz = gz - (float)z0;
                    const float4* v = tpl + (int64_t)k * TD * sD;
                    float4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int xi = x0 + (c & 1), yi = y0 + ((c >> 1) & 1), zi = z0 + (c >> 2);
                        // |y| < 1 strictly => 0 <= x0 <= TW - 2 (and likewise y0, z0): all eight corners are inside the grid,
                        // the reference's zero-padding branch (utils.h:473-500) cannot trigger here
                        const float wgt = ((c & 1) ? fx : 1.f - fx) * (((c >> 1) & 1) ? fy : 1.f - fy) * ((c >> 2) ? fz : 1.f - fz);
                        const float4 q = (stats == 2) ? float4{0.5f, 0.5f, 0.5f, 5.f} : v[zi * sD + yi * sH + xi];
                        s.x += q.x * wgt; s.y += q.y * wgt; s.z += q.z * wgt; s.w += q.w * wgt;
                    }
                    const float alpha = s.w * fade;
                    const float newalpha = acc.w + alpha * stepsize;
                    const float contrib = fminf(newalpha, 1.f) - acc.w;
                    acc.x += s.x * contrib; acc.y += s.y * contrib; acc.z += s.z * contrib; acc.w += contrib;
                    if (newalpha >= 1.f) sat = true;
                }
            }
            t += stepsize;
            rp = {rp.x + rd.x * stepsize, rp.y + rd.y * stepsize, rp.z + rd.z * stepsize};
        }
    }
    if (valid) rayrgba[pix] = acc;
    if (stats && lane == 0) {
        atomicAdd(&g_rm_stats[0], 1ull); atomicAdd(&g_rm_stats[1], (unsigned long long)num);
        atomicAdd(&g_rm_stats[2], st_chunks ? 1ull : 0ull); atomicAdd(&g_rm_stats[3], st_chunks); atomicAdd(&g_rm_stats[4], st_nsub);
        atomicAdd(&g_rm_stats[5], st_steps); atomicAdd(&g_rm_stats[6], st_eval);
        const unsigned long long c_end = __builtin_readcyclecounter();
        atomicAdd(&g_rm_stats[7], c_hit - c_start); atomicAdd(&g_rm_stats[8], c_rebuild); atomicAdd(&g_rm_stats[9], c_end - c_hit - c_rebuild);
        atomicMax(&g_rm_stats[10], c_end - c_start);
    }
}

}  // namespace

extern "C" int primx_compute_raydirs(const float* viewpos, const float* viewrot, const float* focal, const float* princpt,
                                     const float* pixelcoords, float volradius, float* raypos, float* raydir, float* tminmax,
                                     int N, int H, int W, void* stream) {
    PRIMX_REQUIRE(viewpos && viewrot && focal && princpt && raypos && raydir && tminmax, "primx_compute_raydirs: null pointer");
    PRIMX_REQUIRE(N > 0 && H > 0 && W > 0 && volradius > 0.f, "primx_compute_raydirs: empty problem");
    const int64_t n = (int64_t)N * H * W;
    hipLaunchKernelGGL(raydirs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, viewpos, viewrot,
                       focal, princpt, pixelcoords, volradius, raypos, raydir, tminmax, N, H, W);
    PRIMX_CHECK_LAUNCH("primx_compute_raydirs");
    return PRIMX_OK;
}

extern "C" int primx_raymarch(const float* raypos, const float* raydir, const float* tminmax, float stepsize,
                              const float* primpos, const float* primrot, const float* primscale, const float* tplate,
                              float* rayrgba, int N, int H, int W, int K, int TD, int TH, int TW, float fadescale,
                              float fadeexp, void* stream) {
    PRIMX_REQUIRE(raypos && raydir && tminmax && primpos && primrot && primscale && tplate && rayrgba,
                  "primx_raymarch: null pointer");
    PRIMX_REQUIRE(N > 0 && H > 0 && W > 0 && K > 0 && TD > 1 && TH > 1 && TW > 1 && stepsize > 0.f,
                  "primx_raymarch: empty problem / non-positive step");
    static const int stats = [] { const char* e = getenv("PRIMX_RAYMARCH_STATS"); return e ? atoi(e) : 0; }();   // 2: no template fetch
    unsigned long long z[12] = {0}, r[12];
    if (stats) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rm_stats), z, sizeof(z));
    hipLaunchKernelGGL(raymarch_kernel, dim3((W + 15) / 16, (H + 15) / 16, N), dim3(256), 0, (hipStream_t)stream, raypos,
                       raydir, tminmax, stepsize, primpos, primrot, primscale, (const float4*)tplate, (float4*)rayrgba, N, H, W,
                       K, TD, TH, TW, fadescale, fadeexp, stats);
    if (stats) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_rm_stats), sizeof(r));
        const double w = r[0] ? (double)r[0] : 1.0, mw = r[2] ? (double)r[2] : 1.0;
        fprintf(stderr, "raymarch stats: %llu waves, hit list %.1f per wave; %llu marching waves: %.1f chunks, %.1f listed per chunk, "
                        "%.0f steps, %.1f record evaluations per step; cycles per wave: hit list %.0f, rebuilds %.0f (per marching wave), "
                        "marching %.0f (per marching wave), slowest wave %llu\n", r[0], r[1] / w, r[2], r[3] / mw, r[4] / (r[3] ? (double)r[3] : 1.0),
                r[5] / mw, r[6] / (r[5] ? (double)r[5] : 1.0), r[7] / w, r[8] / mw, r[9] / mw, r[10]);
    }
    PRIMX_CHECK_LAUNCH("primx_raymarch");
    return PRIMX_OK;
}
