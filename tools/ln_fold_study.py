#!/usr/bin/env python
"""CPU study (NOT part of the product path): what would folding LayerNorm + modulate into the CONSUMER GEMM's epilogue do to the numbers?

Round-3 review item 3b / round-4 DESIGN.md section 9: the 85 LayerNorm launches of a DDIM step (0.58 ms + their share of the launch
gaps) cannot be removed by an in-kernel hand-over (measured, DESIGN_LOG.md section 10.1); the remaining route is algebraic.  With
mu_r, rho_r the row statistics of the fp32 residual stream x and m = cast16(1 + scale):

    reference (autocast):   y = cast16( cast16( (x - mu) rho m + shift ) W^T + b )
    folded:                 y = cast16( rho ( cast16(x m) W^T  -  mu u ) + v ),      u = m W^T,   v = shift W^T + b      (u, v: fp32, per
                                                                                     (block, timestep, batch entry): two extra GEMV rows)

The producer GEMM's epilogue would write cast16(x m) and per-row partial sums, the consumer's epilogue would apply rho, mu, u, v - no
LayerNorm kernel.  The folded form rounds x m instead of the normalised, modulated value: the same RELATIVE rounding per element, but
the subtraction of mu u cancels, so rows whose mean is large against their spread lose precision.  This script measures that:

  1. operator level: one LN -> modulate -> Linear (D = 1152 -> 1152) on synthetic rows with a controlled |mean| / std ratio and optional
     outlier channels, fp16 and bf16, against float64;
  2. model level: `forward_with_cfg` of the full configs[1] model (28 blocks, synthetic weights of the goldens) with the three folds per
     block + the final layer's, in the oracle's 16-bit emulation, against the golden of the unmodified reference (tests/golden/xl_c2.npz)
     - next to the same emulation WITHOUT the fold.

    python tools/ln_fold_study.py [--skip-model] [--blocks N]

Results of the committed run: DESIGN_LOG.md section 10.4.  The fold was BUILT at the end of round 4 (csrc/gemm.hip "LayerNorm fold",
DESIGN_LOG.md section 10.5) in the centred form: `--as-built` runs the model-level study with the kernels' arithmetic - the centre of a
site is the row mean at the previous site, the statistics come from fp32 partial sums of (x - c), (x - c)^2 over 144-column tiles
(var = E[d^2] - mu'^2), the first LayerNorm of block 0 and the final layer's stay unfolded.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dit_ref, synth  # noqa: E402

r16 = dit_ref._r


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def fold_linear(x, shift, scale, w, b, emulate, eps=1e-6, centre=None):
    """The folded LN -> modulate -> Linear of one site.  x: [B, N, D] fp32; shift / scale: [B, D] (16-bit values); w: [O, D]; b: [O].
    `centre` = None: the uncentred form of the first study (c = 0, exact statistics).  `centre` = a one-element list holding the
    per-row (centre, scale) pair ([B, N, 1] fp32 each) or None: the as-built form (ABI 23) - operand cast16((x - c) rho_p m),
    statistics from fp32 partial sums of d = x - c over 144-column tiles, and the list is updated to this site's (mean, rstd) for
    the next one (None: this is the first site - primx_row_stats supplies the pair)."""
    m = r16(1 + scale, emulate).double().unsqueeze(1)                          # (1 + scale) formed in the 16-bit type, as autocast does
    wd = r16(w, emulate).double()
    u = (m @ wd.t()).float().double()                                          # [B, 1, O] fp32
    v = (r16(shift, emulate).double().unsqueeze(1) @ wd.t() + (0 if b is None else r16(b, emulate).double())).float().double()
    if centre is None:
        xd = x.double()
        mu = xd.mean(-1, keepdim=True)
        rho = 1.0 / torch.sqrt(((xd - mu) ** 2).mean(-1, keepdim=True) + eps)
        mu, rho = mu.float().double(), rho.float().double()                  # fp32 statistics
        a16 = r16((x * m.float()), emulate).double()                           # what the producer's epilogue would store
    else:
        if centre[0] is None:                                                  # (primx_row_stats in front of the first site)
            c = x.mean(-1, keepdim=True)
            rp = (1.0 / torch.sqrt(x.double().var(-1, unbiased=False, keepdim=True) + eps)).float()
        else:
            c, rp = centre[0]
        d = (x - c).float()                                                    # fp32, as the producer's epilogue forms it
        a16 = r16((d * rp) * m.float(), emulate).double()
        D = x.shape[-1]
        dt = d.double().view(*d.shape[:-1], D // 144, 144)
        s1 = dt.sum(-1).float().double().sum(-1, keepdim=True)                 # fp32 partials per 144-column tile, then their sum
        s2 = (dt * dt).sum(-1).float().double().sum(-1, keepdim=True)
        mu = (s1 / D).float().double()                                         # mu' = mean(x - c)
        rho = (1.0 / torch.sqrt(torch.clamp(s2 / D - mu * mu, min=0.0) + eps)).float().double()
        centre[0] = ((c.double() + mu).float(), rho.float())                   # the consumer's column tile 0 writes the next pair
        mu, rho = (rp.double() * mu).float().double(), (rho / rp.double()).float().double()   # the epilogue's (rho_p mu', rho / rho_p)
    acc = a16 @ wd.t()                                                         # fp32-accumulated MFMA (float64 here)
    y = (rho * (acc - mu * u) + v).float()                                     # epilogue arithmetic in fp32
    return r16(y, emulate)


def std_linear(x, shift, scale, w, b, emulate):
    return dit_ref._linear(dit_ref._modulate(dit_ref.layer_norm(x), shift, scale, emulate), w, b, emulate)


def operator_study():
    print("== operator level: LN -> modulate -> Linear(1152 -> 1152), 2048 rows; rel-L2 against float64")
    print(f"{'dtype':6s} {'|mean|/std':>10s} {'outliers':>9s} {'autocast':>10s} {'folded':>10s} {'ratio':>6s}")
    D, O, N = 1152, 1152, 2048
    g = torch.Generator().manual_seed(5)
    w = torch.randn(O, D, generator=g) * D ** -0.5
    b = torch.randn(O, generator=g) * 0.05
    out = []
    for emulate in (torch.float16, torch.bfloat16):
        shift = r16(torch.randn(1, D, generator=g) * 0.3, emulate)
        scale = r16(torch.randn(1, D, generator=g) * 0.3, emulate)
        for ratio in (0.0, 0.5, 2.0, 10.0, 50.0):
            for outl in (False, True):
                x = torch.randn(1, N, D, generator=g) * 3.0
                if outl:                                                      # 1 % of the channels carry 50 x the typical magnitude
                    idx = torch.randperm(D, generator=g)[:D // 100]
                    x[:, :, idx] *= 50.0
                x = x + ratio * x.std(-1, keepdim=True)                       # per-row mean = ratio x the row's spread
                xd = x.double()
                mu = xd.mean(-1, keepdim=True)
                ln = (xd - mu) / torch.sqrt(((xd - mu) ** 2).mean(-1, keepdim=True) + 1e-6)
                ref = (ln * (1 + scale.double()).unsqueeze(1) + shift.double().unsqueeze(1)) @ w.double().t() + b.double()
                e_std = rel(std_linear(x, shift, scale, w, b, emulate), ref)
                e_fold = rel(fold_linear(x, shift, scale, w, b, emulate), ref)
                name = "fp16" if emulate == torch.float16 else "bf16"
                print(f"{name:6s} {ratio:10.1f} {str(outl):>9s} {e_std:10.2e} {e_fold:10.2e} {e_fold / e_std:6.2f}")
                out.append((name, ratio, outl, e_std, e_fold))
    return out


def folded_block(sd, i, x, y, t_emb, H, emulate, centre=None):
    """dit_ref.dit_block with the three LN -> modulate -> Linear sites folded (to_q, qkv, fc1).  `centre`: see fold_linear; in the
    as-built form the first site of block 0 is NOT folded (its LayerNorm stays a launch and the row-mean kernel seeds the centre)."""
    p = f"blocks.{i}."
    mod = dit_ref._linear(F.silu(t_emb), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"], emulate)
    sh_c, sc_c, g_c, sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(9, dim=1)
    B, N, C = x.shape
    dh = C // H
    # cross-attention (attention.py:96-114): q = scale * to_q(.), double scale inside the core
    pc = p + "crossattn."
    s = dh ** -0.5
    if centre is not None and i == 0:
        q0 = std_linear(x, sh_c, sc_c, sd[pc + "to_q.weight"], sd.get(pc + "to_q.bias"), emulate)
        centre[0] = (x.mean(-1, keepdim=True), (1.0 / torch.sqrt(x.double().var(-1, unbiased=False, keepdim=True) + 1e-6)).float())
    else:
        q0 = fold_linear(x, sh_c, sc_c, sd[pc + "to_q.weight"], sd.get(pc + "to_q.bias"), emulate, centre=centre)
    q = r16(s * q0, emulate).reshape(B, N, H, dh)
    M = y.shape[1]
    k = dit_ref._linear(y, sd[pc + "to_k.weight"], sd.get(pc + "to_k.bias"), emulate).reshape(B, M, H, dh)
    v = dit_ref._linear(y, sd[pc + "to_v.weight"], sd.get(pc + "to_v.bias"), emulate).reshape(B, M, H, dh)
    o = r16(dit_ref.attention_core(q, k, v, s), emulate).reshape(B, N, C)
    x = x + r16(g_c.unsqueeze(1) * dit_ref._linear(o, sd[pc + "proj.weight"], sd.get(pc + "proj.bias"), emulate), emulate)
    # self-attention (attention.py:48-59)
    pa = p + "attn."
    qkv = fold_linear(x, sh_a, sc_a, sd[pa + "qkv.weight"], sd.get(pa + "qkv.bias"), emulate, centre=centre).reshape(B, N, 3, H, dh)
    o = r16(dit_ref.attention_core(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dh ** -0.5), emulate).reshape(B, N, C)
    x = x + r16(g_a.unsqueeze(1) * dit_ref._linear(o, sd[pa + "proj.weight"], sd.get(pa + "proj.bias"), emulate), emulate)
    # MLP (models/utils.py:94-101)
    pm = p + "mlp."
    h = r16(F.gelu(fold_linear(x, sh_m, sc_m, sd[pm + "fc1.weight"], sd[pm + "fc1.bias"], emulate, centre=centre), approximate="tanh"), emulate)
    x = x + r16(g_m.unsqueeze(1) * dit_ref._linear(h, sd[pm + "fc2.weight"], sd[pm + "fc2.bias"], emulate), emulate)
    return x


def folded_forward_with_cfg(sd, x, t, y, H, cfg_scale, emulate, stats, as_built=False):
    y_null = sd["null_cond_embedding"].expand_as(y)
    xx, tt, yy = torch.cat([x, x]), torch.cat([t, t]), torch.cat([y, y_null])
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    h = F.linear(xx.float(), sd["x_embedder.weight"], sd["x_embedder.bias"])
    te = dit_ref.timestep_embedding(tt)
    te = F.linear(F.silu(F.linear(te, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])), sd["t_embedder.mlp.2.weight"],
                  sd["t_embedder.mlp.2.bias"])
    centre = [None] if as_built else None
    for i in range(depth):
        r = (h.mean(-1).abs() / h.std(-1)).flatten()
        stats.append((float(r.mean()), float(r.max()), float(h.abs().max())))
        h = folded_block(sd, i, h, yy.float(), te, H, emulate, centre)
    mod = dit_ref._linear(F.silu(te), sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"], emulate)
    shift, scale = mod.chunk(2, dim=1)
    if as_built:                                                               # the final layer's LayerNorm stays a launch
        out = std_linear(h, shift, scale, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"], emulate)
    else:
        out = fold_linear(h, shift, scale, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"], emulate)
    cond, uncond = torch.split(out, len(out) // 2, dim=0)
    return r16(uncond + r16(cfg_scale * r16(cond - uncond, emulate), emulate), emulate)


def model_study(blocks: int, as_built: bool = False):
    from tests.golden.make_golden_xl import HEADS, XL, XL_SEED, xl_inputs
    depth, N, B, stride, x, y = xl_inputs("xl_c2")
    g = np.load(os.path.join(ROOT, "tests", "golden", "xl_c2.npz"))
    t = torch.as_tensor(g["t"])
    nb = blocks or depth
    sd = synth.dit_state_dict(XL_SEED, depth=nb, **XL)
    torch.set_num_threads(os.cpu_count() or 8)
    print(f"== model level: forward_with_cfg, {nb} of {depth} blocks, N = {N}, configs[1] weights / inputs of the golden")
    with torch.no_grad():
        t0 = time.time()
        ref32 = dit_ref.dit_forward_with_cfg(sd, x, t, y, HEADS, 6.0, None)
        print(f"   fp32 oracle: {time.time() - t0:.0f} s" + (f"; vs the reference golden {rel(ref32[:, ::stride], torch.as_tensor(g['forward_cfg'])):.2e}" if nb == depth else ""))
        for emulate in (torch.float16, torch.bfloat16):
            t0 = time.time()
            std = dit_ref.dit_forward_with_cfg(sd, x, t, y, HEADS, 6.0, emulate)
            stats = []
            fold = folded_forward_with_cfg(sd, x, t, y, HEADS, 6.0, emulate, stats, as_built)
            name = "fp16" if emulate == torch.float16 else "bf16"
            print(f"   {name}: autocast emulation vs fp32 {rel(std, ref32):.3e} | folded vs fp32 {rel(fold, ref32):.3e} | folded vs autocast emulation "
                  f"{rel(fold, std):.3e}   ({time.time() - t0:.0f} s)")
            if emulate == torch.float16:
                rm = max(s[1] for s in stats)
                print(f"   residual stream in front of the blocks: |row mean| / row std mean {np.mean([s[0] for s in stats]):.3f}, max {rm:.3f}; "
                      f"largest |x| {max(s[2] for s in stats):.1f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-model", action="store_true")
    ap.add_argument("--blocks", type=int, default=0, help="model level: use only the first n blocks (0 = all 28)")
    ap.add_argument("--as-built", action="store_true", help="model level with the kernels' arithmetic (centred, partial sums); skips the operator table")
    args = ap.parse_args()
    if not args.as_built:
        operator_study()
    if not args.skip_model:
        model_study(args.blocks, args.as_built)


if __name__ == "__main__":
    main()
