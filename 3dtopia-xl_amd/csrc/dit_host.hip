// primx_dit_blocks_fold (ABI 24; the K / V projection riding on the qkv launches: ABI 25): the DiT blocks of one planned, folded forward issued from ONE foreign call.  Host code only: it
// calls the library's own entry points in the order and with the arguments DiT._forward16 (3dtopia-xl_amd/dit.py) issues them, so
// the kernels, their launch shapes and the results are the same - what changes is the host's cost per DDIM step (231 Python +
// ctypes calls of ~21 us -> ~10: DESIGN.md section 8, tools/host_bound_check.py).
// Reference: the block loop of models/dit_crossattn.py:198-199 (DiTBlock.forward :51-58).
#include <stdint.h>

#include "common.h"

#define PRIMX_FAIL(...)               \
    do {                              \
        primx_set_error(__VA_ARGS__); \
        return PRIMX_EINVAL;          \
    } while (0)

extern "C" int primx_dit_blocks_fold(const PrimxDitForwardFold* f, const PrimxDitBlockFold* blocks, void* stream) {
    if (!f || !blocks) PRIMX_FAIL("primx_dit_blocks_fold: null descriptor");
    if (f->dtype != PRIMX_F16 && f->dtype != PRIMX_BF16) PRIMX_FAIL("primx_dit_blocks_fold: dtype must be PRIMX_F16 or PRIMX_BF16");
    if (f->depth <= 0 || f->Be <= 0 || f->N <= 0 || f->D <= 0 || f->H <= 0 || f->dh * f->H != f->D || f->hidden <= 0)
        PRIMX_FAIL("primx_dit_blocks_fold: bad shape (depth %d, Be %d, N %d, D %d, H %d, dh %d, hidden %d)", f->depth, f->Be, f->N, f->D,
                   f->H, f->dh, f->hidden);
    if (f->step < 0 || f->step >= f->n_steps) PRIMX_FAIL("primx_dit_blocks_fold: step %d outside the u / v tables (%d rows)", f->step, f->n_steps);
    if (f->b_from < 0 || f->b_from > f->Be) PRIMX_FAIL("primx_dit_blocks_fold: b_from %d outside [0, %d]", f->b_from, f->Be);
    if (!f->h || !f->xn || !f->att || !f->hid || !f->Qc || !f->Qs || !f->Ks || !f->Vs || !f->mod || !f->center0 || !f->center1 || !f->part)
        PRIMX_FAIL("primx_dit_blocks_fold: null workspace");
    const int dt = f->dtype, Be = f->Be, N = f->N, D = f->D, H = f->H, dh = f->dh, T = Be * N;
    const char* mod = static_cast<const char*>(f->mod);                       // 16-bit elements
    auto ch = [&](int i, int j) -> const void* { return mod + ((int64_t)(i * 9 + j) * D) * 2; };
    float* center[2] = {f->center0, f->center1};
    int side = 0;                                                             // which array the current site's producer used
    const int kq[1] = {PRIMX_HEADS_ROWS}, kqkv[3] = {PRIMX_HEADS_ROWS, PRIMX_HEADS_KROWS, PRIMX_HEADS_VT};
    void* const dq[1] = {f->Qc};
    void* const dqkv[3] = {f->Qs, f->Ks, f->Vs};
    auto uv = [&](const float* base, int n_site, const float*& u, const float*& v) {
        u = base + (int64_t)f->step * n_site;
        v = base + ((int64_t)f->n_steps + f->step) * n_site;
    };
    int rc;
#define PRIMX_DIT_CALL(x)          \
    do {                           \
        rc = (x);                  \
        if (rc != PRIMX_OK) return rc; \
    } while (0)
    // ABI 25: the to_k / to_v projection of the conditioning tokens (attention.py:106-107) issued from here - block 0's as a launch of its
    // own, block i + 1's on the CUs block i's qkv launch leaves idle (primx_linear_heads_fold_pair)
    const int kkv[2] = {PRIMX_HEADS_KROWS, PRIMX_HEADS_VT};
    if (f->kv_A) {
        if (!f->kv_W || f->kv_rows <= 0 || f->kv_rows_per_batch <= 0 || f->kv_rows % f->kv_rows_per_batch || f->kv_K <= 0)
            PRIMX_FAIL("primx_dit_blocks_fold: bad K / V projection (rows %d, rows per batch entry %d, K %d)", f->kv_rows, f->kv_rows_per_batch, f->kv_K);
        for (int i = 0; i < f->depth; ++i)
            if (!blocks[i].Kc || !blocks[i].Vc) PRIMX_FAIL("primx_dit_blocks_fold: block %d: null K / V destination", i);
    }
    const char* const kvW = static_cast<const char*>(f->kv_W);
    const char* const kvB = static_cast<const char*>(f->kv_bias);
#define PRIMX_DIT_KV(i)                                                                                                                      \
    f->kv_A, kvW + (int64_t)(i) * 2 * D * f->kv_K * 2, kvB ? kvB + (int64_t)(i) * 2 * D * 2 : nullptr, f->kv_rows, 2 * D, f->kv_K,         \
        f->kv_rows_per_batch, H, dh, 2, kkv, kvdst, f->nkv_pad_c, 1.0f
    if (f->kv_A) {
        void* const kvdst[2] = {const_cast<void*>(blocks[0].Kc), const_cast<void*>(blocks[0].Vc)};
        PRIMX_DIT_CALL(primx_linear_heads_fold_pair(nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, nullptr,
                                                    nullptr, 0.f, PRIMX_DIT_KV(0), dt, stream));
    }
    for (int i = 0; i < f->depth; ++i) {
        const PrimxDitBlockFold& b = blocks[i];
        if (!b.w_q || !b.w_cproj || !b.w_qkv || !b.w_proj || !b.w_fc1 || !b.w_fc2 || !b.uv_qkv || !b.uv_fc1 || (i > 0 && !b.uv_q))
            PRIMX_FAIL("primx_dit_blocks_fold: block %d: null weight or table", i);
        const float *u, *v;
        // ---- cross-attention to the image tokens (dit_crossattn.py:55, attention.py:96-114)
        if (i == 0) {   // the first LayerNorm of a forward stays a launch; its (mean, rstd) are the first site's (centre, scale)
            PRIMX_DIT_CALL(primx_layernorm_modulate(f->h, ch(0, 0), ch(0, 1), 0, f->xn, dt, T, N, D, f->ln_eps, nullptr, 0, nullptr, 0, stream));
            PRIMX_DIT_CALL(primx_row_stats(f->h, T, D, f->ln_eps, center[0], stream));
            side = 0;
        }
        if (b.uv_q) {
            uv(b.uv_q, D, u, v);
            PRIMX_DIT_CALL(primx_linear_heads_fold(f->xn, b.w_q, T, D, D, N, H, dh, 1, kq, dq, f->nq_pad, f->scale, f->part, u, v, center[side],
                                                   center[side ^ 1], f->ln_eps, dt, b.carry_q, b.carry_q_bytes, stream));
            side ^= 1;
        } else {
            PRIMX_DIT_CALL(primx_linear_heads(f->xn, b.w_q, b.b_q, T, D, D, N, H, dh, 1, kq, dq, 1, 0, f->nq_pad, f->scale, dt, b.carry_q,
                                              b.carry_q_bytes, stream));
        }
        if (f->b_from < Be) {
            if (!b.Kb || !b.Vb || (f->b_from > 0 && (!b.Kc || !b.Vc))) PRIMX_FAIL("primx_dit_blocks_fold: block %d: null cross-attention operand", i);
            PRIMX_DIT_CALL(primx_attention_bcast(f->Qc, f->b_from > 0 ? b.Kc : nullptr, f->b_from > 0 ? b.Vc : nullptr, f->att, Be, H, N, f->nq_pad,
                                                 f->L, f->b_from > 0 ? f->nkv_pad_c : f->nkv_pad_b, dh, f->scale, b.Kb, b.Vb, f->b_from,
                                                 f->nkv_pad_b, dt, stream));
        } else {
            if (!b.Kc || !b.Vc) PRIMX_FAIL("primx_dit_blocks_fold: block %d: null cross-attention operand", i);
            PRIMX_DIT_CALL(primx_attention(f->Qc, b.Kc, b.Vc, f->att, Be, H, N, f->nq_pad, f->L, f->nkv_pad_c, dh, f->scale, dt, stream));
        }
        PRIMX_DIT_CALL(primx_linear_gate_residual_fold(f->att, b.w_cproj, b.b_cproj, ch(i, 2), 0, f->h, T, D, D, N, ch(i, 4), 0, center[side], f->xn,
                                                       f->part, dt, b.carry_cproj, b.carry_cproj_bytes, stream));
        // ---- self-attention over the primitive tokens (dit_crossattn.py:56, attention.py:48-59)
        uv(b.uv_qkv, 3 * D, u, v);
        if (f->kv_A && i + 1 < f->depth) {
            void* const kvdst[2] = {const_cast<void*>(blocks[i + 1].Kc), const_cast<void*>(blocks[i + 1].Vc)};
            PRIMX_DIT_CALL(primx_linear_heads_fold_pair(f->xn, b.w_qkv, T, 3 * D, D, N, H, dh, 3, kqkv, dqkv, f->nq_pad, 1.0f, f->part, u, v,
                                                        center[side], center[side ^ 1], f->ln_eps, PRIMX_DIT_KV(i + 1), dt, stream));
        } else {
            PRIMX_DIT_CALL(primx_linear_heads_fold(f->xn, b.w_qkv, T, 3 * D, D, N, H, dh, 3, kqkv, dqkv, f->nq_pad, 1.0f, f->part, u, v, center[side],
                                                   center[side ^ 1], f->ln_eps, dt, nullptr, 0, stream));
        }
        side ^= 1;
        PRIMX_DIT_CALL(primx_attention(f->Qs, f->Ks, f->Vs, f->att, Be, H, N, f->nq_pad, N, f->nq_pad, dh, f->scale, dt, stream));
        PRIMX_DIT_CALL(primx_linear_gate_residual_fold(f->att, b.w_proj, b.b_proj, ch(i, 5), 0, f->h, T, D, D, N, ch(i, 7), 0, center[side], f->xn,
                                                       f->part, dt, nullptr, 0, stream));
        // ---- MLP (dit_crossattn.py:57, models/utils.py:94-101)
        uv(b.uv_fc1, f->hidden, u, v);
        PRIMX_DIT_CALL(primx_linear_fold(f->xn, b.w_fc1, f->hid, T, f->hidden, D, PRIMX_ACT_GELU_TANH, f->part, u, v, center[side], center[side ^ 1],
                                         f->ln_eps, dt, b.carry_fc1, b.carry_fc1_bytes, stream));
        side ^= 1;
        if (i + 1 < f->depth) {   // producer of the next block's to_q site: the next LayerNorm's scale is block i + 1's chunk 1
            PRIMX_DIT_CALL(primx_linear_gate_residual_fold(f->hid, b.w_fc2, b.b_fc2, ch(i, 8), 0, f->h, T, D, f->hidden, N, ch(i + 1, 1), 0,
                                                           center[side], f->xn, f->part, dt, b.carry_fc2, b.carry_fc2_bytes, stream));
        } else {                  // the final layer's LayerNorm stays a launch (its Linear has 8 x out_channels columns)
            PRIMX_DIT_CALL(primx_linear_gate_residual_ln(f->hid, b.w_fc2, b.b_fc2, ch(i, 8), 0, f->h, T, D, f->hidden, N, ch(f->depth, 0),
                                                         ch(f->depth, 1), 0, f->xn, f->ln_eps, nullptr, 0, dt, nullptr, 0, stream));
        }
    }
#undef PRIMX_DIT_KV
#undef PRIMX_DIT_CALL
    return PRIMX_OK;
}
