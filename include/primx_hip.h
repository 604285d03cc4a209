/*
 * primx_hip.h - C ABI of libprimx_hip.so: the MI355X (gfx950) hot path of 3DTopia-XL's
 * DDIM denoising loop (PrimX DiT) and 3D-VAE decode.
 *
 * Boundary rules
 *   - plain C: device pointers, sizes, a HIP stream handle passed as void*; no torch types.
 *   - every entry point returns 0 on success or a negative PRIMX_E* code; the message of the
 *     last failure on the calling thread is available from primx_last_error().
 *   - buffers are owned by the caller (PyTorch's caching allocator in the Python host); kernels
 *     keep no pointers after return and never allocate; the library keeps NO per-thread or global
 *     state between calls that changes what a later call does (ABI 21: the cache-prefetch ranges
 *     that GEMM / LayerNorm launches can carry are explicit arguments of the carrying call).
 *     Work is enqueued on `stream` and is asynchronous with respect to the host.
 *   - `dtype` selects the 16-bit storage/MFMA-input type of activations and weights:
 *     PRIMX_F16 or PRIMX_BF16.  Accumulation is always fp32.
 *
 * Each entry cites the reference code it replaces (paths relative to the 3DTopia-XL repo).
 * The reference has no native FFI on this path (it is PyTorch + xFormers); its one functional
 * seam is xformers.ops.memory_efficient_attention (models/attention.py:17,54,109), and the
 * convention mirrored here is the one of its CUDA extensions (caller-allocated outputs,
 * dva/mvp/extensions/mvpraymarch/mvpraymarch.py:132-138), except that the current stream is
 * honoured instead of stream 0 (mvpraymarch.cpp:121).
 */
#ifndef PRIMX_HIP_H
#define PRIMX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRIMX_ABI_VERSION 26

/* dtype codes */
#define PRIMX_F32 0
#define PRIMX_F16 1
#define PRIMX_BF16 2

/* error codes */
#define PRIMX_OK 0
#define PRIMX_EINVAL (-1)   /* bad argument (null pointer, unsupported shape/dtype) */
#define PRIMX_ELAUNCH (-2)  /* hipGetLastError() after a launch was not hipSuccess */

/* activation codes for primx_linear */
#define PRIMX_ACT_NONE 0
#define PRIMX_ACT_GELU_TANH 1
#define PRIMX_ACT_GELU_ERF 2  /* nn.GELU() exact form (DINOv2 Mlp) */

/* head-layout kinds for primx_linear_heads / primx_pack_heads */
#define PRIMX_HEADS_ROWS 0 /* [B, H, n_pad, DP]  token-major rows, head dim zero-padded to DP   */
#define PRIMX_HEADS_VT 1   /* [B, H, DP, n_pad]  transposed, keys permuted inside 16-groups     */
#define PRIMX_HEADS_KROWS 2 /* [B, H, n_pad, DP+8] rows with the padded stride of the attention kernel's LDS image: a
                             * 64-key tile is ONE contiguous block that LDS-DMA copies verbatim (K operand only)  */

int primx_abi_version(void);
const char* primx_last_error(void);
/* Name of the GEMM kernel instantiation that the last primx_linear / primx_linear_gate_residual / primx_linear_heads /
 * primx_linear_residual / primx_conv3d_k3 / primx_convtranspose_k2s2 call on THIS thread launched, spelled as rocprofv3
 * prints it (e.g. "gemm144l_dma_kernel<1, 1>"); "" before the first call.  Measurement plumbing only (bench.py tags its
 * per-launch HIP-event timings with it): no counterpart in the reference.  ABI 19. */
const char* primx_last_gemm_kernel(void);

/* Padded head dim used by the attention layouts: smallest multiple of 16 >= dh (72 -> 80). */
int primx_padded_head_dim(int dh);

/* ----------------------------------------------------------------------------------------------
 * Row kernels of the DiT block
 * -------------------------------------------------------------------------------------------- */

/* out[r, :] = cast( LN(x[r, :]) * (1 + scale[b, :]) + shift[b, :] ),  b = r / rows_per_batch.
 * LN has no affine, fp32 statistics, eps as given.  `shift`/`scale` are 16-bit vectors of
 * length D taken at element stride `mod_stride` between batch entries (they are chunks of the
 * adaLN output).  (1 + scale) is rounded to the 16-bit type first, as autocast does.
 * Replaces nn.LayerNorm(elementwise_affine=False, eps=1e-6) + modulate():
 * models/dit_crossattn.py:32-36,55-57,67,76 and models/utils.py:19-20.   D even, D <= 2048.
 * pf0 / pf1 (each (pointer, bytes > 0) or (NULL, 0)): byte ranges that this launch also pulls into the caches - its grid gets
 * extra leading workgroups that load one word per 128-byte line (D % 128 == 0 fast path; other shapes ignore the ranges).  Meant
 * for the weights of the GEMMs that follow; results are unaffected.  The ranges must stay allocated until the launch has run. */
int primx_layernorm_modulate(const float* x, const void* shift, const void* scale, int64_t mod_stride,
                             void* out, int dtype, int rows, int rows_per_batch, int D, float eps,
                             const void* pf0, int64_t pf0_bytes, const void* pf1, int64_t pf1_bytes, void* stream);

/* emb[b, :] = [cos(t_b * f_k) | sin(t_b * f_k)], k < dim/2.  `freqs` (dim/2 floats, device) is the
 * table f_k = exp(-ln(max_period) * k / (dim/2)) that the reference also evaluates on the HOST before
 * moving it to the device (models/utils.py:51-54); the product and the sin/cos run here in fp32.
 * Replaces TimestepEmbedder.timestep_embedding, models/utils.py:40-59 (dim even). */
int primx_timestep_embedding(const int64_t* t, const float* freqs, float* emb, int B, int dim, void* stream);

/* PointEmbed features of the DiTAdditivePosEmb variant (models/dit_crossattn.py:80-108, 283-285): for token t with
 * point p = x[t*row_stride + 1 .. 3] writes feat[t*feat_stride + :] = [sin(p_d * freqs[k]) (d-major, 3F), cos(...) (3F), p (3)]
 * (freqs[k] = 2^k * pi, the non-zero entries of the reference's block-diagonal `basis` buffer). */
/* ViT token assembly of the DINOv2 conditioner (dinov2/models/vision_transformer.py:218-236): fp32
 * out[b, 0] = cls + pos[0];  out[b, 1 + r] = reg[r] (r < R, no positional term);  out[b, 1 + R + i] = patches[b, i] + pos[1 + i]. */
int primx_vit_tokens(const float* patches, const float* cls, const float* pos, const float* reg, float* out, int B, int np,
                     int R, int D, void* stream);

int primx_point_features(const float* x, int64_t row_stride, const float* freqs, float* feat, int64_t feat_stride, int T,
                         int F, void* stream);

/* Touch every 128-byte line of [ptr, ptr + bytes): a cache prefetch (Infinity Cache / L2), no result.  Enqueue it on a SIDE
 * stream while a compute-bound kernel (attention, LayerNorm) runs, for the weights of the GEMM that follows.  No counterpart
 * in the reference (its weights are whatever the caches hold); results are unaffected.  ABI 19. */
int primx_prefetch(const void* ptr, int64_t bytes, void* stream);
/* out = cast16( silu(in) ) elementwise.  The SiLU in front of every adaLN Linear
 * (models/dit_crossattn.py:40-43,69-72) producing the 16-bit GEMM operand. */
int primx_silu_cast(const float* in, void* out, int dtype, int64_t n, void* stream);

/* out = cast16(in) elementwise: the fp32 -> fp16/bf16 cast autocast applies to a Linear's input
 * (here: the conditioning tokens y in front of to_k / to_v, attention.py:106-107). */
int primx_cast16(const float* in, void* out, int dtype, int64_t n, void* stream);

/* Small fp32 linear: out[m, n] = act_out( sum_k in[m, k] * W[n, k] + bias[n] ), W in nn.Linear
 * (out, in) layout, everything fp32 (these layers run outside autocast in the reference).
 * act_out: 0 none, 1 SiLU.  Used for x_embedder (models/dit_crossattn.py:141,191) and the
 * TimestepEmbedder MLP (models/utils.py:33-37).  `out2` (NULL or a second [M, N] buffer, M > 8): receives the same rows -
 * forward_with_cfg embeds cat([x, x]) (dit_crossattn.py:205), i.e. the same tokens into both halves of the residual stream. */
int primx_linear_f32(const float* in, const float* W, const float* bias, float* out, float* out2, int M, int N, int K,
                     int act_out, void* stream);

/* ----------------------------------------------------------------------------------------------
 * MFMA GEMMs with fused epilogues.  A: [M, K] 16-bit row-major activations; W: [N, K] 16-bit,
 * nn.Linear (out, in) layout; bias: [N] 16-bit or NULL.  K % 8 == 0.  fp32 accumulate.
 * `prefetch` / `prefetch_bytes` ((pointer, bytes > 0) or (NULL, 0)) of primx_linear, primx_linear_gate_residual[_ln] and
 * primx_linear_heads: a byte range - the weights of a GEMM one or two launches ahead - that THIS launch pulls towards the caches.
 * The compute waves of the loader-wave kernels never use their vector-memory queue inside the k-loop, so each of them requests
 * one word per 128-byte line of the range in front of the loop (at most 1024 lines per workgroup: 33.5 MB at 256 workgroups,
 * longer ranges are cut) and the lines travel HBM -> Infinity Cache while the loop runs from L2.  Launches that take a kernel
 * without loader waves ignore the range; results are unaffected; the range must stay allocated until the launch has run.
 * No counterpart in the reference (its weights are whatever the caches hold).
 * -------------------------------------------------------------------------------------------- */

/* out[M, N] (16-bit) = out_scale * act(A W^T + bias), each stage rounded to the 16-bit type as
 * autocast does (Linear output, activation output, scaled output).  out_scale == 1 skips the
 * last rounding.  Replaces nn.Linear under autocast: Mlp.fc1 + GELU(tanh) (models/utils.py:87-96,
 * dit_crossattn.py:38), adaLN Linear (dit_crossattn.py:40-43), FinalLayer.linear (:68). */
int primx_linear(const void* A, const void* W, const void* bias, void* out, int M, int N, int K, int dtype,
                 int act, float out_scale, const void* prefetch, int64_t prefetch_bytes, void* stream);

/* x[m, :] += cast16( gate[b, :] * cast16(A W^T + bias)[m, :] ),  b = m / rows_per_batch, x fp32.
 * Replaces the projection Linear + gated residual `x = x + gate.unsqueeze(1) * branch`:
 * attention.py:56,111 / models/utils.py:98 with dit_crossattn.py:55-57. */
int primx_linear_gate_residual(const void* A, const void* W, const void* bias, const void* gate,
                               int64_t gate_stride, float* x, int M, int N, int K, int rows_per_batch,
                               int dtype, const void* prefetch, int64_t prefetch_bytes, void* stream);

/* primx_linear_gate_residual FOLLOWED BY primx_layernorm_modulate of the updated rows, as one call:
 *   x[m, :] += cast16(gate[b, :] * cast16(A W^T + bias)[m, :]);   ln_out[m, :] = cast16( LN(x[m, :]) * (1 + ln_scale[b, :]) + ln_shift[b, :] )
 * - in a DiT block every gated residual add is followed by the LayerNorm + modulate of the next branch (or of the next block, or of
 * the final layer): models/dit_crossattn.py:55-57,76.  Results are bit-identical to the two separate calls.  Where the shape allows
 * (N == 1152, M a multiple of 128 * 8 up to its last ragged block, the loader-wave 128 x 144 kernel) the LayerNorm runs in the
 * TAIL of the GEMM kernel: the column-tile workgroups of a 128-row block count themselves in on `sync`, wait for each other and
 * normalise 16 rows each from the L2 they share - one dependent launch less per LayerNorm (85 per DDIM step).  Otherwise the
 * library launches the two kernels.  `sync`: device memory, two 32-bit words per 128-row block (sync_words >= 2 * ceil(M / 128)),
 * zeroed once by the caller; every launch leaves them zero; launches that use the same words must be ordered by the stream.
 * sync == NULL always takes the two-launch route.  primx_last_gemm_kernel() tells which route ran ("gemm144l_dma_kernel<dt, 5>" =
 * fused).  ABI 21. */
int primx_linear_gate_residual_ln(const void* A, const void* W, const void* bias, const void* gate, int64_t gate_stride,
                                  float* x, int M, int N, int K, int rows_per_batch, const void* ln_shift, const void* ln_scale,
                                  int64_t ln_mod_stride, void* ln_out, float ln_eps, void* sync, int64_t sync_words, int dtype,
                                  const void* prefetch, int64_t prefetch_bytes, void* stream);
/* Number of in-kernel waits of primx_linear_gate_residual_ln that gave up (~1 s each) since the library was loaded: 0 unless the
 * dispatch-order assumption of the fused route is violated on this machine (then PRIMX_LN_FUSE=0 selects the two-launch route).
 * Synchronises with the device.  ABI 21. */
int primx_ln_sync_timeouts(void);

/* ---- The LayerNorm fold (ABI 22; range-safe operand since ABI 23).  Every LayerNorm + modulate of a DiT block sits between a
 * gated residual add and a Linear (models/dit_crossattn.py:55-57).  With the row statistics mu, rho of the fp32 residual stream x,
 * m = cast16(1 + scale), any per-row centre c and any per-row scale rho_p > 0:
 *     reference:  y = cast16( cast16( (x - mu) rho m + shift ) W^T + b )
 *     folded:     y = cast16( (rho / rho_p) cast16((x - c) rho_p m) W^T - rho (mu - c) u + v ),   u = m W^T,  v = shift W^T + b   (fp32 rows)
 * so the gate-residual GEMM in front of the LayerNorm (the PRODUCER) can store the 16-bit operand a16 = cast16((x - c) rho_p m) and
 * per-column-tile partial sums of (x - c), (x - c)^2, and the Linear behind it (the CONSUMER: to_q, qkv, fc1) can finish mu, rho
 * from the partials and apply them with u, v in its epilogue - no LayerNorm launch, no second pass over the fp32 rows.
 * `center` is an array of (c, rho_p) PAIRS, [rows][2] fp32: the row's (mean, rstd) at the previous LayerNorm site - primx_row_stats
 * in front of the first site, afterwards what the previous consumer wrote to its `center_out`.  With them the operand is the
 * LayerNorm output up to what ONE gated branch changes mean and spread by: the accuracy of the reference's rounding (which rounds
 * the normalised value; tools/ln_fold_study.py) and - ABI 23 - its RANGE: |a16| = O(|1 + scale|) whatever the magnitude or spread
 * of the residual stream, in PRIMX_F16 as in PRIMX_BF16 (ABI 22 stored cast16((x - c) m), which carried the row's spread).
 * A consumer reads `center` (all its column tiles need rho_p) and writes the next pair to `center_out`, which must be a different
 * array: callers alternate two.  u, v depend on the timestep only (primx_linear_f32out, once per planned sampling loop).
 * Shapes: N of the producer = K of the consumer = a multiple of 144, at most 1152; everything else returns PRIMX_EINVAL and the
 * caller keeps primx_layernorm_modulate. */

/* out[m, n] (fp32) = sum_k A[m, k] W[n, k] + (m >= bias_from_row ? bias[n] : 0): fp32 rows out of 16-bit operands.  The fold's
 * u rows (A = cast16(1 + scale) of each planned timestep, no bias) and v rows (A = shift, bias = the Linear's) in one launch. */
int primx_linear_f32out(const void* A, const void* W, const void* bias, float* out, int M, int N, int K, int bias_from_row,
                        int dtype, void* stream);

/* ABI 26.  Many primx_linear_f32out problems with the same M, K, bias_from_row and dtype from ONE launch: the fold's u / v rows of every
 * (block, site) of a planned sampling loop (83 problems at DiT-XL: 592 MB of weights streamed once instead of 83 latency-bound launches).
 * `probs` lives in DEVICE memory: problem i = out[M, N] (fp32, 16-byte aligned rows: N % 32 == 0) = A[M, K] W[N, K]^T (+ bias, may be NULL,
 * for the rows >= bias_from_row); first_wg = the sum of ceil(N / 128) over the problems in front of it (ascending from 0), total_wg the sum
 * over all.  Same products in the same order for every output whatever M and the problem count. */
typedef struct PrimxF32outProblem {
    const void* A;
    const void* W;
    const void* bias;
    float* out;
    int N;
    int first_wg;
} PrimxF32outProblem;
int primx_linear_f32out_group(const PrimxF32outProblem* probs, int n_probs, int total_wg, int M, int K, int bias_from_row, int dtype,
                              void* stream);

/* stats[r] = (mean(x[r, :]), 1 / sqrt(var(x[r, :]) + eps)) in fp32, [rows][2]: the (c, rho_p) pair of the first folded site of a
 * forward - the statistics primx_layernorm_modulate uses for the same rows (D % 4 == 0).  ABI 23 (replaces primx_row_mean). */
int primx_row_stats(const float* x, int rows, int D, float eps, float* stats, void* stream);

/* primx_linear_gate_residual that is also the PRODUCER of the LayerNorm site behind it ((c, rho_p) = center[m][0 / 1]):
 *   x[m, :] += cast16(gate[b, :] * cast16(A W^T + bias)[m, :]);   a16_out[m, :] = cast16( (x[m, :] - c) * rho_p * cast16(1 + next_scale[b, :]) );
 *   part_out[m, t, 0 / 1] = sum over the 144 columns of tile t of (x - c), (x - c)^2      (t < N / 144, fixed summation order).
 * next_scale: the 16-bit scale vectors of the NEXT LayerNorm's modulate, element stride next_mod_stride between batch entries. */
int primx_linear_gate_residual_fold(const void* A, const void* W, const void* bias, const void* gate, int64_t gate_stride,
                                    float* x, int M, int N, int K, int rows_per_batch, const void* next_scale,
                                    int64_t next_mod_stride, const float* center, void* a16_out, float* part_out, int dtype,
                                    const void* prefetch, int64_t prefetch_bytes, void* stream);

/* primx_linear_heads (n_rep = 1) as the CONSUMER of a folded site: A = the producer's a16, `part` its partial sums (K / 144 per
 * row), `center` the (c, rho_p) pairs the producer used, u / v = fp32 vectors of N columns (16-byte aligned; the Linear's bias is
 * part of v), eps = the LayerNorm's.  The workgroups of column tile 0 write center_out[m] = (c + mean(x - c), rho): the next
 * producer's pair (center_out != center). */
int primx_linear_heads_fold(const void* A, const void* W, int M, int N, int K, int rows_per_batch, int heads, int dh, int n_seg,
                            const int* kind, void* const* dst, int n_pad, float scale0, const float* part, const float* u,
                            const float* v, const float* center, float* center_out, float eps, int dtype, const void* prefetch,
                            int64_t prefetch_bytes, void* stream);

/* ABI 25.  primx_linear_heads_fold (problem 0: the arguments above without the prefetch range) and primx_linear_heads (problem 1:
 * A2 ... scale0_2, n_rep = 1, no prefetch range) from ONE launch where both map onto the 256 x 288 tile's heads epilogue and fit
 * one round of the chip together (problem 0's workgroups a multiple of 8, at least PRIMX_GEMM_BIGHEADS_MIN; problem 0 + problem 1
 * <= 256): problem 1's tiles run on the CUs problem 0 leaves idle - the self-attention qkv projection of a DiT block at T = 4096
 * (attention.py:48-54: 192 tiles) carries the to_k / to_v projection of the NEXT block's conditioning tokens (attention.py:106-107:
 * 48 tiles).  Results are those of the two calls, bit for bit; outside the rule the two launches are made one after the other.
 * A == NULL: problem 1 alone, on the tile kernel it would have ridden (its bits do not depend on whether it rode). */
int primx_linear_heads_fold_pair(const void* A, const void* W, int M, int N, int K, int rows_per_batch, int heads, int dh, int n_seg,
                                 const int* kind, void* const* dst, int n_pad, float scale0, const float* part, const float* u,
                                 const float* v, const float* center, float* center_out, float eps,
                                 const void* A2, const void* W2, const void* bias2, int M2, int N2, int K2, int rows_per_batch2,
                                 int heads2, int dh2, int n_seg2, const int* kind2, void* const* dst2, int n_pad2, float scale0_2,
                                 int dtype, void* stream);

/* primx_linear (out_scale = 1) as the CONSUMER of a folded site (fc1 + GELU):
 * out = act(cast16((rho / rho_p) a16 W^T - rho mu' u + v)); center / center_out as above. */
int primx_linear_fold(const void* A, const void* W, void* out, int M, int N, int K, int act, const float* part, const float* u,
                      const float* v, const float* center, float* center_out, float eps, int dtype, const void* prefetch,
                      int64_t prefetch_bytes, void* stream);

/* Projection whose output columns are `n_rep` repetitions of `n_seg` groups of (heads * dh) features
 * (N = n_rep * n_seg * heads * dh), each group written straight into an attention operand layout (see
 * PRIMX_HEADS_*): group s of repetition r goes to batch entries [r * rep_batches, (r+1) * rep_batches) of dst[s] with kind[s]; rows of
 * batch b (= m / rows_per_batch) go to [b, h, m % rows_per_batch, :].  Group 0 is multiplied by `scale0`
 * (after rounding) - the cross-attention `self.scale * to_q(q)` (attention.py:105).  Pad rows/cols of the
 * destinations are never written (callers zero them once).  n_rep > 1 batches the SAME projection of several
 * DiT blocks over one input: the to_k/to_v projections of all 28 blocks read the same conditioning tokens.
 * Replaces qkv Linear + reshape + unbind (attention.py:50-52) and to_q/to_k/to_v + reshape
 * (attention.py:105-107). */
int primx_linear_heads(const void* A, const void* W, const void* bias, int M, int N, int K, int rows_per_batch,
                       int heads, int dh, int n_seg, const int* kind, void* const* dst, int n_rep,
                       int rep_batches, int n_pad, float scale0, int dtype, const void* prefetch, int64_t prefetch_bytes,
                       void* stream);

/* ----------------------------------------------------------------------------------------------
 * Attention (flash-style, fp32 online softmax, MFMA 32x32x16)
 * -------------------------------------------------------------------------------------------- */

/* out[b, q, h*dh + d] = sum_k softmax_k( scale * <Q[b,h,q,:], K[b,h,k,:]> ) V[b,h,k,d]
 * Qp: [B, H, nq_pad, DP] ROWS layout (nq_pad % 128 == 0); Kp: [B, H, nkv_pad, DP+8] KROWS layout;
 * Vt: [B, H, DP, nkv_pad] VT layout (nkv_pad % 64 == 0).  out: [B, nq, H*dh] 16-bit.  dh in {32, 64, 72}.
 * Keys >= nkv are masked.  For dh == DP (32, 64) the kernel overwrites their scores; for dh < DP (72 -> 80)
 * the mask is carried BY THE OPERANDS, which the caller prepares once (the pads are never written by any
 * kernel): Qp[.., q, dh] = 1 for every row, Kp[.., key, dh] = -30000 for key in [nkv, nkv_pad) and 0 for valid
 * keys, Vt[.., dh, pos(key)] = 1 for valid keys (the all-ones row: the PV MFMA accumulates the softmax
 * denominator in output row dh), every other pad entry of Qp/Kp/Vt zero - a padded key then scores -30000
 * through the MFMA itself and its probability underflows to exactly 0, with no mask or row-sum code in the
 * softmax (ops.alloc_heads does this).  When DP - dh >= 3 the caller additionally sets Kp[.., key, dh+1] =
 * Kp[.., key, dh+2] = 1 for EVERY key row (Qp stays 0 there): the kernel keeps the running row max, split into two
 * 16-bit halves and negated, in its register copy of Q's columns dh+1 / dh+2, so the MFMA itself subtracts it.
 * Small problems (the VAE mid-block attention: nq, nkv <= 64, dh == 32) may come in COMPACT buffers, nq_pad == nkv_pad == 64;
 * they run on a one-wave-per-problem kernel that reads the operands straight from memory (no 256-row workgroup, no
 * padding to 128 / 256 tokens).
 * Replaces xformers.ops.memory_efficient_attention(q, k, v) (attention.py:54,109). */
int primx_attention(const void* Qp, const void* Kp, const void* Vt, void* out, int B, int H, int nq, int nq_pad,
                    int nkv, int nkv_pad, int dh, float scale, int dtype, void* stream);
/* primx_attention in which the batch entries b >= b_from attend to nkv COPIES OF ONE key / value row - the unconditional half of
 * `DiT.forward_with_cfg`, whose conditioning is `null_cond_embedding.expand_as(y)` (models/dit_crossattn.py:204-209): L identical
 * tokens, hence L identical rows of to_k(y_null) / to_v(y_null) (models/attention.py:106-107).  Those entries share ONE operand
 * entry Kb [1, H, nkv_pad_b, DP + 8] / Vb [1, H, DP, nkv_pad_b] in the layouts of Kp / Vt that holds the sequence once: keys
 * [0, 64) = the row, and - when nkv % 64 != 0 - keys [64, 64 + nkv % 64) = the row again with the pad keys behind them masked
 * like any pad keys (primx_linear_heads on 64 + nkv % 64 identical rows writes exactly this).  The kernel walks all nkv keys
 * as always and only maps the tile address (ragged last tile -> tile 1, every other tile -> tile 0): the same tile contents
 * and arithmetic as with the expanded operands, without projecting, storing and re-reading nkv copies.  Kp / Vt hold the
 * entries [0, b_from) (may be NULL when b_from == 0); b_from == B is primx_attention.  Needs nkv >= 64; not available on the
 * one-wave 64-token kernel.  ABI 20. */
int primx_attention_bcast(const void* Qp, const void* Kp, const void* Vt, void* out, int B, int H, int nq, int nq_pad, int nkv,
                          int nkv_pad, int dh, float scale, const void* Kb, const void* Vb, int b_from, int nkv_pad_b, int dtype,
                          void* stream);

/* ----------------------------------------------------------------------------------------------
 * The DiT blocks of one planned, folded forward from ONE call (ABI 24)
 * -------------------------------------------------------------------------------------------- */

/* Weights and per-block operands of DiTBlock i (models/dit_crossattn.py:26-58): 16-bit (out, in) matrices and biases of
 * CrossAttention.to_q / .proj (models/attention.py:83-87), Attention.qkv / .proj (:37-39), Mlp.fc1 / .fc2 (models/utils.py:87-101);
 * the block's cross-attention operands as primx_linear_heads wrote them (Kc / Vc: the entries with their own conditioning tokens,
 * Kb / Vb: the broadcast entry of primx_attention_bcast or NULL); the LayerNorm fold's per-timestep rows of the block's three sites
 * (primx_linear_f32out: fp32 [2][n_steps][N_site], u rows then v rows; uv_q == NULL: the site keeps its LayerNorm launch - block 0).
 * `carry_*`: the cache-prefetch range (pointer, bytes; (NULL, 0) = none) that the launch of that GEMM carries. */
typedef struct PrimxDitBlockFold {
    const void *w_q, *b_q, *w_cproj, *b_cproj, *w_qkv, *w_proj, *b_proj, *w_fc1, *w_fc2, *b_fc2;
    const void *Kc, *Vc, *Kb, *Vb;
    const float *uv_q, *uv_qkv, *uv_fc1;
    const void *carry_q, *carry_cproj, *carry_fc1, *carry_fc2;
    int64_t carry_q_bytes, carry_cproj_bytes, carry_fc1_bytes, carry_fc2_bytes;
} PrimxDitBlockFold;

/* One forward's shapes, workspaces and tables.  Be = effective batch (2 B under classifier-free guidance), T = Be * N token rows.
 * h: fp32 residual stream [T, D] (in / out); xn: 16-bit [T, D] (LayerNorm output / folded operand); att: 16-bit [Be, N, D];
 * hid: 16-bit [T, hidden]; Qc, Qs: ROWS operands [Be, H, nq_pad, DP]; Ks: KROWS [Be, H, nq_pad, DP + 8]; Vs: VT [Be, H, DP, nq_pad];
 * mod: this forward's adaLN row (16-bit, depth * 9 D + 2 D elements: per block shift / scale / gate of the cross-attention,
 * self-attention and MLP branch, then the final layer's shift / scale - dit_crossattn.py:54,69-75), shared by all batch entries;
 * center0 / center1: the fold's two (centre, scale) arrays [T][2] fp32; part: [T][D / 144][2] fp32; step: this forward's row of the
 * u / v tables, n_steps their row count.  b_from: first batch entry that attends to the broadcast conditioning entry (== Be: none);
 * L: conditioning tokens, nkv_pad_c / nkv_pad_b: padded key counts of Kc / Kb. */
typedef struct PrimxDitForwardFold {
    int dtype, Be, N, D, H, dh, hidden, depth, L, nq_pad, nkv_pad_c, nkv_pad_b, b_from, step, n_steps;
    float ln_eps, scale;
    float* h;
    void *xn, *att, *hid, *Qc, *Qs, *Ks, *Vs;
    const void* mod;
    float *center0, *center1, *part;
    /* ABI 25: the cross-attention K / V projection of the conditioning tokens done by this call (kv_A != NULL): kv_A = 16-bit
     * [kv_rows, kv_K] conditioning rows (kv_rows_per_batch per batch entry, a multiple of 256), kv_W / kv_bias = [depth * 2 D, kv_K] /
     * [depth * 2 D]: per block [to_k; to_v] (attention.py:106-107); block i's result goes to its Kc / Vc (KROWS / VT, nkv_pad_c keys).
     * Block 0's projection is a launch of its own in front of the blocks, block i + 1's rides on block i's qkv launch
     * (primx_linear_heads_fold_pair).  kv_A == NULL: the caller has filled Kc / Vc. */
    const void *kv_A, *kv_W, *kv_bias;
    int kv_rows, kv_rows_per_batch, kv_K;
} PrimxDitForwardFold;

/* The `depth` DiT blocks of a forward whose LayerNorms are folded (every call below is one of this header's entry points, issued
 * in the order and with the arguments `DiT._forward16` of the Python host issues them - same kernels, bit-identical results; the
 * host makes ONE foreign call per forward instead of 8 per block: 231 -> 10 per DDIM step at depth 28).  Per block:
 *   block 0 only: primx_layernorm_modulate + primx_row_stats + primx_linear_heads (to_q);  other blocks: primx_linear_heads_fold (to_q)
 *   primx_attention[_bcast] (cross)  ->  primx_linear_gate_residual_fold (cross proj)  ->  primx_linear_heads_fold (qkv; with
 *   kv_A: primx_linear_heads_fold_pair carrying the next block's to_k / to_v)
 *   ->  primx_attention (self)  ->  primx_linear_gate_residual_fold (proj)  ->  primx_linear_fold (fc1 + GELU-tanh)
 *   ->  primx_linear_gate_residual_fold (fc2; last block: primx_linear_gate_residual_ln with the final layer's shift / scale).
 * Replaces the loop `for block in self.blocks: x = block(x, y, c)` (models/dit_crossattn.py:198-199) of a planned sampling loop. */
int primx_dit_blocks_fold(const PrimxDitForwardFold* f, const PrimxDitBlockFold* blocks, void* stream);

/* Gather a [B, M, H, dh] tensor with arbitrary element strides (the xFormers BMHK operand, e.g.
 * a view into the fused qkv buffer) into an attention operand layout. */
int primx_pack_heads(const void* src, int64_t sb, int64_t sm, int64_t sh, void* dst, int kind, int B, int M,
                     int H, int dh, int m_pad, int dtype, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Classifier-free guidance + the diffusion update
 * -------------------------------------------------------------------------------------------- */

/* out[b] = uncond[b] + s * (cond[b] - uncond[b]) with cond = in[0:B], uncond = in[B:2B], every
 * operation rounded to the storage type (fp16/bf16 under autocast, fp32 otherwise).
 * Replaces DiT.forward_with_cfg's combine, models/dit_crossattn.py:210-213.  n = elements per half. */
int primx_cfg_combine(const void* in, void* out, int dtype, int64_t n, float s, void* stream);

/* One reverse-diffusion update, fully fused.  coef: device table [n_steps, 16] fp32 built by the
 * host (sampler.py step_coefficients), `step` selects the row.
 *   mean_type 0 eps / 1 x0 / 2 v;  var_type 0 fixed-small / 1 fixed-large / 2 learned / 3 learned-range
 *   ancestral 0: DDIM  (gaussian_diffusion.py:531-578);  1: p_sample (gaussian_diffusion.py:394-435)
 * model_out: [B, n_tok, c_out] with c_out = C or 2C (variance channels second), dtype out_dtype.
 * noise may be NULL when it would be multiplied by zero.  Writes sample and pred_xstart (fp32).
 * Replaces p_mean_variance + _predict_xstart_from_z_and_v + _predict_eps_from_xstart + the DDIM
 * / ancestral formulas: gaussian_diffusion.py:255-356,394-435,531-578,880-892. */
int primx_diffusion_step(const float* x, const void* model_out, int out_dtype, int64_t n_rows, int C, int c_out,
                         const float* coef, int step, int mean_type, int var_type, int ancestral, int clip,
                         const float* noise, float* sample, float* pred_xstart, void* stream);

/* ----------------------------------------------------------------------------------------------
 * 3D-VAE decoder (models/vae3d_dib.py:330-387,437-440).  Activations are channels-last 16-bit:
 * [P, V, C] with V = S^3 voxels in (z, y, x) raster order.
 * -------------------------------------------------------------------------------------------- */

/* GroupNorm(groups, eps, affine) + optional SiLU over one primitive's [V, C] block, fp32
 * statistics; in/out 16-bit channels-last.  (vae3d_dib.py:109-112,131-139,366,383-384) */
int primx_groupnorm_silu(const void* in, const float* gamma, const float* beta, void* out, int P, int V, int C,
                         int groups, float eps, int silu, int dtype, void* stream);

/* 3x3x3 convolution, stride 1, zero padding 1, on an S^3 grid, channels-last, as an implicit GEMM
 * on MFMA: out[p, v, co] = ((bias[co] + sum_{tap, ci} in[p, v + tap, ci] * Wk[co, tap*Cin + ci]) + res[p, v, co])
 * * res_scale.  Wk is the 16-bit weight re-laid by the host as [Cout, Kpad] with k = tap*Cin + ci
 * (tap = (dz*3 + dy)*3 + dx), zero-padded to Kpad (a multiple of 64).  res may be NULL.
 * Cin a power of two >= 8.
 * Replaces nn.Conv3d(k=3, p=1) in ResnetBlock incl. the skip `(x + shortcut(res)) * skip_scale`
 * (vae3d_dib.py:110,113,137-143) and, with a flipped/transposed weight, the output
 * ConvTranspose3d(k=3, s=1, p=1) (vae3d_dib.py:367,385). */
int primx_conv3d_k3(const void* in, const void* Wk, const void* bias, const void* res, float res_scale,
                    void* out, int P, int S, int Cin, int Cout, int Kpad, int dtype, void* stream);

/* The same convolution for the decoder's 4^3 stage (S = 4, Cin = 256, Cout a multiple of 256: the eight ResnetBlock
 * convolutions of mid_block and up_blocks[0], vae3d_dib.py:62-75 / 251-261) with the activations held in registers as
 * MFMA operand fragments and the taps formed by fragment selection + DPP row shifts (csrc/conv3.hip).  The weight is
 * pre-packed ONCE by primx_conv3d_s4_pack from the [Cout, 27*256] layout of primx_conv3d_k3 (Kpad == 6912) into the
 * LDS images of the kernel's [256 cout][64 k] tiles (same size: Cout * 6912 elements; Wp must not alias Wk).
 * Same result formula and rounding as primx_conv3d_k3. */
int primx_conv3d_s4_pack(const void* Wk, void* Wp, int Cout, int dtype, void* stream);
int primx_conv3d_s4_packed(const void* in, const void* Wp, const void* bias, const void* res, float res_scale, void* out,
                           int P, int Cout, int dtype, void* stream);

/* The same convolution for S = 8, Cin = 256, Cout = 32 (conv1 of up_blocks[1].nets[0], vae3d_dib.py:62-75 / 262-270):
 * one workgroup per primitive, each wave keeps one input z-plane in registers and scatters every tap's product into an
 * fp32 image of the output in LDS (csrc/conv3s8.hip).  primx_conv3d_s8_pack re-lays the [32, 6912] weight of
 * primx_conv3d_k3 into the kernel's LDS tile images (same size; Wp must not alias Wk).  Same result formula as
 * primx_conv3d_k3; the fp32 summation order differs (per tap over all channels, then over taps). */
int primx_conv3d_s8_pack(const void* Wk, const void* Wsc, void* Wp, int dtype, void* stream);
int primx_conv3d_s8_packed(const void* in, const void* Wp, const void* bias, const void* res, float res_scale, void* out,
                           int P, int dtype, void* stream);
/* primx_conv3d_s8_pack: Wp holds 27 blocks of 8192 elements; with Wsc != NULL (the [32, 256] weight of the ResnetBlock's
 * 1x1 shortcut, vae3d_dib.py:124-125) a 28th block is written and Wp must hold 28.
 * primx_conv3d_s8_fused: the whole front of up_blocks[1].nets[0] (vae3d_dib.py:109-110, 124-125) on the RAW upsample
 * output: norm1 (GroupNorm, 32 groups of 8 channels) from the partial sums `part` [P, 16, 32, 2] of
 * primx_convtranspose_s4_packed (shift = up_bias[8 g]) + SiLU applied in registers, conv1 -> out [P, 512, 32], and
 * shortcut(in_raw) = Wsc x in_raw + sc_bias -> sc_out [P, 512, 32] (what primx_linear_residual(…, res = NULL) gives).
 * Replaces primx_groupnorm_silu (1.6 GB of traffic at 2048 primitives) + primx_conv3d_s8_packed + primx_linear_residual.
 * SiLU uses v_rcp_f32 (1 ulp) where primx_groupnorm_silu divides; both round the result to 16 bits. */
int primx_conv3d_s8_fused(const void* in_raw, const void* Wp28, const void* bias, const float* part, const void* up_bias,
                          const float* gamma, const float* beta, float eps, const void* sc_bias, void* out, void* sc_out, int P,
                          int dtype, void* stream);

/* The same convolution for S = 8, Cin = 32, Cout = 32 or <= 16, optionally with the preceding GroupNorm (ONE channel per
 * group: per-(primitive, channel) statistics over the 512 voxels) + SiLU applied to the input inside the kernel
 * (gamma/beta fp32 [32], both NULL = plain convolution of `in`): conv2 of up_blocks[1].nets[0], conv1/conv2 of nets[1] and
 * norm_out + conv_out (vae3d_dib.py:62-75, 262-270, 366-367, 383-385).  The primitive's activations live in LDS as a
 * zero-haloed volume next to the whole weight (csrc/conv3s8c32.hip).  primx_conv3d_s8c32_pack re-lays the [Cout, Kpad]
 * weight of primx_conv3d_k3 (Kpad >= 864) into the kernel's image: 27 * 32 * 32 elements for Cout = 32, 27 * 16 * 32
 * for Cout <= 16.  out = ((conv(silu(gn(in))) + bias) + res) * res_scale, one rounding; the normalised activations are
 * rounded to 16 bits before the convolution exactly as primx_groupnorm_silu rounds them. */
int primx_conv3d_s8c32_pack(const void* Wk, void* Wp, int Cout, int Kpad, int dtype, void* stream);
int primx_conv3d_s8c32_packed(const void* in, const void* Wp, const void* bias, const float* gamma, const float* beta, float eps,
                              const void* res, float res_scale, void* out, int P, int Cout, int dtype, void* stream);

/* primx_convtranspose_k2s2 for S = 4, Cin = Cout = 256 (UpBlock.upsample of up_blocks[0], vae3d_dib.py:250-261),
 * weight-stationary: a workgroup keeps one tap's [256, 256] matrix in LDS and walks primitives (csrc/convt.hip).
 * primx_convtranspose_s4_pack re-lays Wt [8*256, 256] (row = tap*256 + co, as primx_convtranspose_k2s2 takes it) into the
 * kernel's LDS images (same size; must not alias).  `part` (may be NULL): [P, 16, 32, 2] fp32 - for every primitive, 16
 * partial pairs (sum of (x - shift), sum of (x - shift)^2) per group of 8 output channels over disjoint 32-voxel pieces of
 * the ROUNDED 16-bit output, shift = (float) bias[8 * group]: added up in index order they are the statistics of the
 * GroupNorm(32) that follows (ResnetBlock.norm1, vae3d_dib.py:109); primx_conv3d_s8_packed consumes them. */
int primx_convtranspose_s4_pack(const void* Wt, void* Wp, int dtype, void* stream);
int primx_convtranspose_s4_packed(const void* in, const void* Wp, const void* bias, void* out, float* part, int P, int dtype,
                                  void* stream);

/* out[M, N] (16-bit) = ((A W^T + bias) + res) * scale with no intermediate rounding; res may be NULL.
 * The 1x1 shortcut conv (vae3d_dib.py:124-125) and VolumeAttention's proj + `(x + res) * skip_scale`
 * (vae3d_dib.py:43-46) on channels-last activations. */
int primx_linear_residual(const void* A, const void* W, const void* bias, const void* res, float scale, void* out,
                          int M, int N, int K, int dtype, void* stream);

/* conv_in: Conv3d(1 -> Cout, k3, p1) on the fp32 latent grid after post_quant_conv's scalar affine
 * z = a * x + b (vae3d_dib.py:344,373,429,438).  in: [P, S^3] fp32; W: [Cout, 27] fp32; out 16-bit CL. */
int primx_conv_in(const float* in, float pq_scale, float pq_bias, const float* W, const float* bias, void* out,
                  int P, int S, int Cout, int dtype, void* stream);

/* ConvTranspose3d(C -> C, k=2, s=2): S^3 -> (2S)^3, no tap overlap; Wt: [8*Cout, Cin] 16-bit, row = tap*Cout + co,
 * tap = (dz*2 + dy)*2 + dx.
 * (vae3d_dib.py:251,264-265) */
int primx_convtranspose_k2s2(const void* in, const void* Wt, const void* bias, void* out, int P, int S, int Cin,
                             int Cout, int dtype, void* stream);

/* Final layout change + inverse normalisation: channels-last 16-bit [P, V, C] -> fp32 [P, C, V]
 * with channel 0 divided by sdf_div and the others mapped (x + 1) / 2 when `denorm` != 0
 * (inference.py:345-346); denorm == 0 gives the raw VAE.decode output. */
int primx_vae_output(const void* in, float* out, int P, int V, int C, int denorm, float sdf_div, int dtype,
                     void* stream);

/* Latent de-normalisation + split after sampling: v = x / nf * std[c] + mean[c] (each op rounded to fp32, as
 * torch does); channels [0, n_srt) -> srt [rows, n_srt], the rest -> z [rows, C - n_srt] (the VAE latent).
 * Replaces inference.py:328-332 / app.py:119-123. */
int primx_latent_denorm(const float* x, const float* mean, const float* stdv, float nf, float* srt, float* z,
                        int64_t rows, int C, int n_srt, void* stream);

/* ----------------------------------------------------------------------------------------------
 * PrimSDF field query (models/primsdf.py:52-109; inference.py:106-116, 180-193)
 * -------------------------------------------------------------------------------------------- */

/* pts [n, 3] fp32 query points; srt [P, 4] = (scale, x, y, z) per primitive (`srt_param`); feat [P, C * S^3]
 * (`feat_param`, channel-major [C][z][y][x]); lin [S] = torch.linspace(-1, 1, S) (the local grid of the nearest-voxel
 * fallback); out [n, C] = [sdf, clip(tex, 0, 1) x 3, clip(mat, 0, 1) x 2].  eval_fill != 0 applies the inference-time
 * fill of points no primitive covers (primsdf.py:78-100). */
int primx_primsdf_query(const float* pts, const float* srt, const float* feat, const float* lin, float* out, int n, int P,
                        int S, int C, int eval_fill, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Primitive ray marcher, forward (dva/ray_marcher.py:142-229; dva/mvp/extensions/{utils,mvpraymarch})
 * -------------------------------------------------------------------------------------------- */

/* viewpos [N,3], viewrot [N,3,3], focal [N,2], princpt [N,2], pixelcoords [N,H,W,2] or NULL (then (w, h)) ->
 * raypos / raydir [N,H,W,3], tminmax [N,H,W,2] (slab test against the [-1,1]^3 volume, positions / volradius).
 * Replaces compute_raydirs (utils/utils_kernel.cu:15-56). */
int primx_compute_raydirs(const float* viewpos, const float* viewrot, const float* focal, const float* princpt,
                          const float* pixelcoords, float volradius, float* raypos, float* raydir, float* tminmax, int N,
                          int H, int W, void* stream);

/* primpos [N,K,3], primrot [N,K,3,3], primscale [N,K,3] (inverse half extents), tplate [N,K,TD,TH,TW,4] channels-last
 * RGBA -> rayrgba [N,H,W,4].  Replaces mvpraymarch(..., algo=0, chlast=True, warp=None, usebvh="fixedorder") forward
 * (mvpraymarch/mvpraymarch_subset_kernel.h:9-93 with PrimTransfSRT, PrimSamplerTW<false>, PrimAccumAdditive). */
int primx_raymarch(const float* raypos, const float* raydir, const float* tminmax, float stepsize, const float* primpos,
                   const float* primrot, const float* primscale, const float* tplate, float* rayrgba, int N, int H, int W,
                   int K, int TD, int TH, int TW, float fadescale, float fadeexp, void* stream);

/* ----------------------------------------------------------------------------------------------
 * The reference's fp32 call path: DiT.forward(x, t, y) with the signature defaults precision_dtype=float32,
 * enable_amp=False (models/dit_crossattn.py:184) and `precision: tf32` of the CLI (inference.py:239-247).
 * gfx950 has no TF32: these run EXACT fp32 on v_mfma_f32_32x32x2_f32 (csrc/fp32.hip).
 * -------------------------------------------------------------------------------------------- */

/* nn.Linear in fp32 on the matrix cores.  gate == NULL: out[M, N] = act(A W^T + bias) * out_scale.
 * gate != NULL: out[m, n] += gate[(m / rows_per_batch) * gate_stride + n] * (A W^T + bias)[m, n], in place -
 * `x = x + gate.unsqueeze(1) * branch` (models/dit_crossattn.py:55-57).  A [M, K], W [N, K] row-major, K % 4 == 0,
 * 16-byte aligned.  Replaces F.linear outside autocast: models/attention.py:37-39,83-87, models/utils.py:87-91,
 * models/dit_crossattn.py:40-43,68-72. */
int primx_gemm_f32(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int act,
                   float out_scale, const float* gate, int64_t gate_stride, int rows_per_batch, void* stream);

/* xformers.ops.memory_efficient_attention(q, k, v) semantics for fp32 operands given as strided [B, M, H, dh] views
 * (element strides {batch, token, head}; last dim contiguous) - e.g. the unbind() views of the fused qkv buffer:
 * out [B, Nq, H, dh] contiguous = softmax(q k^T * scale) v, fp32 online softmax.  dh <= 128.
 * Replaces models/attention.py:54,109 when autocast is off. */
int primx_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int Nq, int Nkv,
                        int dh, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, float scale,
                        void* stream);

/* out[r, :] = LN(x[r, :]) * (1 + scale[b, :]) + shift[b, :] in fp32 (no affine; b = r / rows_per_batch; shift/scale
 * at element stride mod_stride between batch entries).  models/dit_crossattn.py:32-36,55-57,67,76 outside autocast. */
int primx_layernorm_modulate_f32(const float* x, const float* shift, const float* scale, int64_t mod_stride, float* out,
                                 int rows, int rows_per_batch, int D, float eps, void* stream);

/* out = x * sigmoid(x), fp32: the nn.SiLU in front of every adaLN Linear (models/dit_crossattn.py:40-43,69-72). */
int primx_silu_f32(const float* in, float* out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PRIMX_HIP_H */
