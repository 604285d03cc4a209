"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the DINOv2 conditioner forward.

Follows the reference's vendored implementation (plain fp32 torch, state_dict keys of DinoVisionTransformer):
  * prepare_tokens_with_masks        models/conditioner/dinov2/models/vision_transformer.py:218-236
  * interpolate_pos_encoding         .../vision_transformer.py:188-216
  * PatchEmbed.forward               .../layers/patch_embed.py:68-81   (conv k = s = patch, flatten, no norm)
  * Block.forward (eval)             .../layers/block.py:90-115        x + ls1(attn(norm1 x)); x + ls2(mlp(norm2 x))
  * Attention.forward                .../layers/attention.py:56-70     q scaled by dh^-0.5, softmax, proj
  * Mlp.forward                      .../layers/mlp.py                 fc1 -> GELU (exact) -> fc2
  * LayerScale.forward               .../layers/layer_scale.py         x * gamma
  * forward_features                 .../vision_transformer.py:266-283 final LayerNorm (eps 1e-6), split cls / reg / patches
  * Dinov2Wrapper.forward (tail)     models/conditioner/image_dinov2.py:56-61   cat(cls, patches)
Pinned by tests/golden/dinov2.npz (outputs of the real vendored code, tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def pos_encoding(sd: Dict[str, Tensor], npatch: int, w: int, h: int, patch: int, offset: float = 0.0,
                 antialias: bool = True) -> Tensor:
    pe = sd["pos_embed"].float()
    N = pe.shape[1] - 1
    if npatch == N and w == h:
        return pe
    dim = pe.shape[-1]
    w0, h0 = w // patch + offset, h // patch + offset
    s = math.sqrt(N)
    grid = F.interpolate(pe[:, 1:].reshape(1, int(s), int(s), dim).permute(0, 3, 1, 2),
                         scale_factor=(float(w0) / s, float(h0) / s), mode="bicubic", antialias=antialias)
    return torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)], dim=1)


def forward_features(sd: Dict[str, Tensor], x: Tensor, patch: int, num_heads: int) -> Dict[str, Tensor]:
    B, _, w, h = x.shape
    t = F.conv2d(x.float(), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], dim=1)
    t = t + pos_encoding(sd, t.shape[1] - 1, w, h, patch)
    R = 0
    if "register_tokens" in sd:
        R = sd["register_tokens"].shape[1]
        t = torch.cat([t[:, :1], sd["register_tokens"].expand(B, -1, -1), t[:, 1:]], dim=1)
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    D = t.shape[-1]
    dh = D // num_heads
    for i in range(depth):
        p = f"blocks.{i}."
        n = F.layer_norm(t, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        qkv = F.linear(n, sd[p + "attn.qkv.weight"], sd.get(p + "attn.qkv.bias")).reshape(B, -1, 3, num_heads, dh)
        q, k, v = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        a = ((q * dh ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1) @ v
        a = F.linear(a.transpose(1, 2).reshape(B, -1, D), sd[p + "attn.proj.weight"], sd.get(p + "attn.proj.bias"))
        t = t + a * sd[p + "ls1.gamma"]
        n = F.layer_norm(t, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        m = F.linear(F.gelu(F.linear(n, sd[p + "mlp.fc1.weight"], sd.get(p + "mlp.fc1.bias"))),
                     sd[p + "mlp.fc2.weight"], sd.get(p + "mlp.fc2.bias"))
        t = t + m * sd[p + "ls2.gamma"]
    xn = F.layer_norm(t, (D,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    return {"x_norm_clstoken": xn[:, 0], "x_norm_regtokens": xn[:, 1:R + 1], "x_norm_patchtokens": xn[:, R + 1:],
            "x_prenorm": t}


def conditioner_tokens(sd: Dict[str, Tensor], x: Tensor, patch: int, num_heads: int) -> Tensor:
    out = forward_features(sd, x, patch, num_heads)
    return torch.cat([out["x_norm_clstoken"].unsqueeze(1), out["x_norm_patchtokens"]], dim=1)
