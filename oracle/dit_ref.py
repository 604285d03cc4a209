"""Functional CPU restatement of the PrimX DiT forward (TEST INFRASTRUCTURE - see oracle/__init__.py).

``dit_forward(sd, x, t, y, num_heads)`` follows models/dit_crossattn.py:184-202 op by op on a
plain ``state_dict``.  Two modes:

* ``emulate=None``  - pure fp32, the mode pinned against the imported reference (golden vectors).
* ``emulate=torch.float16 | torch.bfloat16`` - the same arithmetic with a round-to-16-bit inserted at
  every point where ``torch.autocast`` makes the reference round (dit_crossattn.py:197; SURVEY.md
  section 7 "Mixed-precision topology"): Linear inputs/weights/outputs, GELU output, (1 + scale), the
  scaled cross-attention q, gate * branch, attention output, CFG arithmetic.  Contractions themselves
  accumulate in fp32/fp64 here, as the MFMA path does.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _r(x: Tensor, emulate: Optional[torch.dtype]) -> Tensor:
    return x if emulate is None else x.to(emulate).to(torch.float32)


def _linear(x: Tensor, w: Tensor, b: Optional[Tensor], emulate) -> Tensor:
    """nn.Linear; under autocast the input, weight and bias are cast and the output is 16-bit."""
    if emulate is None:
        return F.linear(x, w, b)
    y = F.linear(_r(x, emulate).double(), _r(w, emulate).double(), None if b is None else _r(b, emulate).double())
    return _r(y.float(), emulate)


def timestep_embedding(t: Tensor, dim: int = 256, max_period: float = 10000.0) -> Tensor:
    """models/utils.py:40-59: [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(P) k / (dim/2)), fp32."""
    half = dim // 2
    k = torch.arange(half, dtype=torch.float32)
    freqs = torch.exp(-math.log(max_period) * k / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


ATTN_DTYPE = torch.float64  # bench.py's cpu_baseline leg sets fp32 (the reference's own CPU precision)


def attention_core(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tensor:
    """xformers.ops.memory_efficient_attention semantics on [B, M, H, K] (attention.py:54,109):
    softmax(q k^T * scale) v, no mask, fp32 softmax.  Evaluated in float64 for a clean reference."""
    qd, kd, vd = (z.to(ATTN_DTYPE).permute(0, 2, 1, 3) for z in (q, k, v))
    logits = qd @ kd.transpose(-1, -2) * scale
    p = torch.softmax(logits, dim=-1)
    return (p @ vd).permute(0, 2, 1, 3).float()


def layer_norm(x: Tensor, eps: float = 1e-6) -> Tensor:
    """nn.LayerNorm(elementwise_affine=False) (dit_crossattn.py:32-36) - fp32 even under autocast."""
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def _modulate(xn: Tensor, shift: Tensor, scale: Tensor, emulate) -> Tensor:
    """models/utils.py:19-20.  shift/scale are 16-bit under autocast, so (1 + scale) rounds."""
    one_plus = _r(1 + scale, emulate)
    return xn * one_plus.unsqueeze(1) + shift.unsqueeze(1)


def self_attention(sd: Dict[str, Tensor], p: str, x: Tensor, H: int, emulate) -> Tensor:
    """MemEffAttention._forward (attention.py:48-59)."""
    B, N, C = x.shape
    qkv = _linear(x, sd[p + "qkv.weight"], sd.get(p + "qkv.bias"), emulate).reshape(B, N, 3, H, C // H)
    q, k, v = qkv.unbind(2)
    o = _r(attention_core(q, k, v, (C // H) ** -0.5), emulate).reshape(B, N, C)
    return _linear(o, sd[p + "proj.weight"], sd.get(p + "proj.bias"), emulate)


def cross_attention(sd: Dict[str, Tensor], p: str, x: Tensor, y: Tensor, H: int, emulate) -> Tensor:
    """MemEffCrossAttention._forward (attention.py:96-114): q is scaled by head_dim**-0.5 explicitly
    AND the attention core applies its default head_dim**-0.5 again."""
    B, N, C = x.shape
    M = y.shape[1]
    dh = C // H
    s = dh ** -0.5
    q = _r(s * _linear(x, sd[p + "to_q.weight"], sd.get(p + "to_q.bias"), emulate), emulate).reshape(B, N, H, dh)
    k = _linear(y, sd[p + "to_k.weight"], sd.get(p + "to_k.bias"), emulate).reshape(B, M, H, dh)
    v = _linear(y, sd[p + "to_v.weight"], sd.get(p + "to_v.bias"), emulate).reshape(B, M, H, dh)
    o = _r(attention_core(q, k, v, s), emulate).reshape(B, N, C)
    return _linear(o, sd[p + "proj.weight"], sd.get(p + "proj.bias"), emulate)


def mlp(sd: Dict[str, Tensor], p: str, x: Tensor, emulate) -> Tensor:
    """Mlp.forward with GELU(tanh) (models/utils.py:94-101, dit_crossattn.py:38)."""
    h = _linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"], emulate)
    h = _r(F.gelu(h, approximate="tanh"), emulate)
    return _linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"], emulate)


def dit_block(sd: Dict[str, Tensor], i: int, x: Tensor, y: Tensor, t_emb: Tensor, H: int, emulate) -> Tensor:
    """DiTBlock._forward (dit_crossattn.py:51-58)."""
    p = f"blocks.{i}."
    mod = _linear(F.silu(t_emb), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"], emulate)
    sh_c, sc_c, g_c, sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(9, dim=1)
    x = x + _r(g_c.unsqueeze(1) * cross_attention(sd, p + "crossattn.", _modulate(layer_norm(x), sh_c, sc_c, emulate),
                                                  y, H, emulate), emulate)
    x = x + _r(g_a.unsqueeze(1) * self_attention(sd, p + "attn.", _modulate(layer_norm(x), sh_a, sc_a, emulate), H,
                                                 emulate), emulate)
    x = x + _r(g_m.unsqueeze(1) * mlp(sd, p + "mlp.", _modulate(layer_norm(x), sh_m, sc_m, emulate), emulate), emulate)
    return x


def point_embed(sd: Dict[str, Tensor], point: Tensor) -> Tensor:
    """PointEmbed.forward (dit_crossattn.py:99-108): projections = point @ basis (3 x 24, block diagonal 2^k pi),
    features = [sin, cos, point] -> Linear(51, D).  fp32, outside autocast."""
    proj = torch.einsum("bnd,de->bne", point, sd["point_emb.basis"])
    feat = torch.cat([proj.sin(), proj.cos(), point], dim=2)
    return F.linear(feat, sd["point_emb.mlp.weight"], sd["point_emb.mlp.bias"])


def dit_forward(sd: Dict[str, Tensor], x: Tensor, t: Tensor, y: Tensor, num_heads: int,
                emulate: Optional[torch.dtype] = None) -> Tensor:
    """DiT.forward in eval mode (dit_crossattn.py:184-202).  Returns fp32 holding 16-bit-representable
    values when ``emulate`` is set."""
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    h = F.linear(x.float(), sd["x_embedder.weight"], sd["x_embedder.bias"])  # fp32, outside autocast
    if "point_emb.mlp.weight" in sd:   # DiTAdditivePosEmb (dit_crossattn.py:283-285) + PointEmbed (80-108)
        h = h + point_embed(sd, x.float()[:, :, 1:4])
    te = timestep_embedding(t)
    te = F.linear(F.silu(F.linear(te, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])),
                  sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])
    for i in range(depth):
        h = dit_block(sd, i, h, y.float(), te, num_heads, emulate)
    # FinalLayer.forward (dit_crossattn.py:74-78)
    mod = _linear(F.silu(te), sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"],
                  emulate)
    shift, scale = mod.chunk(2, dim=1)
    h = _modulate(layer_norm(h), shift, scale, emulate)
    return _linear(h, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"], emulate)


def dit_forward_with_cfg(sd: Dict[str, Tensor], x: Tensor, t: Tensor, y: Tensor, num_heads: int, cfg_scale: float,
                         emulate: Optional[torch.dtype] = None) -> Tensor:
    """DiT.forward_with_cfg (dit_crossattn.py:204-213): batch doubled with the null embedding expanded
    over all condition positions; combine over ALL output channels; returns the B-sized half."""
    y_null = sd["null_cond_embedding"].expand_as(y)
    out = dit_forward(sd, torch.cat([x, x]), torch.cat([t, t]), torch.cat([y, y_null]), num_heads, emulate)
    cond, uncond = torch.split(out, len(out) // 2, dim=0)
    return _r(uncond + _r(cfg_scale * _r(cond - uncond, emulate), emulate), emulate)
