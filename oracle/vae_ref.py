"""Functional CPU restatement of ``VAE.decode`` (TEST INFRASTRUCTURE - see oracle/__init__.py).

Follows models/vae3d_dib.py:437-440 -> Decoder.forward (369-387) -> MidBlock (220-226), UpBlock
(259-267), ResnetBlock (128-145), VolumeAttention (34-48) on a plain ``state_dict``; fp32.
``emulate`` rounds activations to a 16-bit type where the HIP decoder stores them (after every conv /
norm / attention op) so the two can be compared tightly; ``None`` is the pure-fp32 reference mode.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .dit_ref import attention_core

Tensor = torch.Tensor
SKIP = 0.5 ** 0.5  # skip_scale = sqrt(0.5) (vae3d_dib.py:338,400)


def _r(x, emulate):
    return x if emulate is None else x.to(emulate).to(torch.float32)


def _gn(sd, p, x, groups=32, eps=1e-5):
    return F.group_norm(x, min(groups, x.shape[1]), sd[p + "weight"], sd[p + "bias"], eps)


def resnet_block(sd: Dict[str, Tensor], p: str, x: Tensor, emulate) -> Tensor:
    """ResnetBlock.forward, resample='default' (vae3d_dib.py:128-145)."""
    res = x
    h = _r(F.silu(_gn(sd, p + "norm1.", x)), emulate)
    h = _r(F.conv3d(h, _r(sd[p + "conv1.weight"], emulate), _r(sd[p + "conv1.bias"], emulate), padding=1), emulate)
    h = _r(F.silu(_gn(sd, p + "norm2.", h)), emulate)
    h = F.conv3d(h, _r(sd[p + "conv2.weight"], emulate), _r(sd[p + "conv2.bias"], emulate), padding=1)
    if p + "shortcut.weight" in sd:
        res = _r(F.conv3d(res, _r(sd[p + "shortcut.weight"], emulate), _r(sd[p + "shortcut.bias"], emulate)), emulate)
    return _r((h + res) * SKIP, emulate)


def volume_attention(sd: Dict[str, Tensor], p: str, x: Tensor, heads: int, emulate) -> Tensor:
    """VolumeAttention.forward (vae3d_dib.py:34-48) with MemEffAttention(qkv_bias=False) (attention.py:48-59)."""
    B, C, H, W, D = x.shape
    res = x
    h = _r(_gn(sd, p + "norm.", x), emulate)
    h = h.permute(0, 2, 3, 4, 1).reshape(B, -1, C)
    qkv = _r(F.linear(h, _r(sd[p + "attn.qkv.weight"], emulate), None), emulate).reshape(B, -1, 3, heads, C // heads)
    q, k, v = qkv.unbind(2)
    o = _r(attention_core(q, k, v, (C // heads) ** -0.5), emulate).reshape(B, -1, C)
    o = _r(F.linear(o, _r(sd[p + "attn.proj.weight"], emulate), _r(sd[p + "attn.proj.bias"], emulate)), emulate)
    o = o.reshape(B, H, W, D, C).permute(0, 4, 1, 2, 3)
    return _r((o + res) * SKIP, emulate)


def vae_decode(sd: Dict[str, Tensor], z: Tensor, up_channels: Sequence[int] = (256, 32), layers_per_block: int = 2,
               heads: int = 8, emulate: Optional[torch.dtype] = None) -> Tensor:
    """z: (P, latent_channels, 4, 4, 4) -> (P, out_channels, 8, 8, 8)."""
    x = F.conv3d(z.float(), sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])      # vae3d_dib.py:438
    d = "decoder."
    x = _r(F.conv3d(x, sd[d + "conv_in.weight"], sd[d + "conv_in.bias"], padding=1), emulate)  # :373
    m = d + "mid_block."
    x = resnet_block(sd, m + "nets.0.", x, emulate)                                        # :221
    x = volume_attention(sd, m + "attns.0.", x, heads, emulate)                           # :224
    x = resnet_block(sd, m + "nets.1.", x, emulate)                                        # :225
    for i in range(len(up_channels)):                                                      # :379-380
        u = d + f"up_blocks.{i}."
        for j in range(layers_per_block):
            x = resnet_block(sd, u + f"nets.{j}.", x, emulate)
        if u + "upsample.weight" in sd:                                                    # ConvTranspose3d k2 s2 (:264-265)
            x = _r(F.conv_transpose3d(x, _r(sd[u + "upsample.weight"], emulate), _r(sd[u + "upsample.bias"], emulate),
                                      stride=2), emulate)
    x = _r(F.silu(_gn(sd, d + "norm_out.", x)), emulate)                                  # :383-384
    return F.conv_transpose3d(x, _r(sd[d + "conv_out.weight"], emulate), _r(sd[d + "conv_out.bias"], emulate),
                              stride=1, padding=1)                                         # :385


def denormalise_decoded(dec: Tensor) -> Tensor:
    """inference.py:345-346: SDF channel / 5, colour+material channels (x + 1) / 2."""
    out = dec.clone()
    out[:, 0:1] /= 5.0
    out[:, 1:] = (out[:, 1:] + 1) / 2.0
    return out
