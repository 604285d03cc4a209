"""DINOv2 ViT conditioner forward on the MI355X kernels - SURVEY.md section 8(f) row N1, the step right before the
sampling loop (`y = conditioner.encoder(image)`, inference.py:317).

Mirrors the vendored reference implementation for the shipped configuration (`dinov2_vitb14_reg`,
configs/inference_dit.yml:49-50; models/conditioner/dinov2/hub/backbones.py:20-72): `DinoVisionTransformer`
(models/conditioner/dinov2/models/vision_transformer.py) with the SAME state_dict keys (`cls_token`, `pos_embed`,
`register_tokens`, `patch_embed.proj.*` (the reference's hacked copy has no `mask_token`), `blocks.N.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,mlp.fc1,
mlp.fc2,ls2.gamma}`, `norm.*`), `forward_features(x)` returning the reference's dict, and `conditioner_tokens(x)` =
what `Dinov2Wrapper.forward` returns after its preprocessing (cat of the cls token and the patch tokens,
image_dinov2.py:56-61).  The modulated variant (`modulation_dim`), masking, chunked blocks and SwiGLU are not part of
the shipped model and raise.

Numerics: fp32 token / residual stream, LayerNorm in fp32, 16-bit GEMM operands and attention (the kernels of the DiT
path: `primx_linear*`, `primx_attention` at dh = 64, `primx_layernorm_modulate` with (gamma - 1, beta) as the
modulation), LayerScale fused into the projection epilogue.  The reference runs this network in fp32 once per image
(0.3 TFLOP, against 158 TFLOP for the 25-step loop); the stated tolerance vs the fp32 reference is rel-L2 5e-3.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
from torch import nn

from . import ops
from ._lib import ACT_GELU_ERF
from .attention import _c16


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True, proj_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden, bias=True):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden, bias=bias)
        self.fc2 = nn.Linear(hidden, dim, bias=bias)


class _LayerScale(nn.Module):
    def __init__(self, dim, init_values):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _Block(nn.Module):
    """Parameter container of dinov2/layers/block.py:Block (norm1, attn, ls1, norm2, mlp, ls2)."""

    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias, init_values):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads, qkv_bias, proj_bias)
        self.ls1 = _LayerScale(dim, init_values)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), ffn_bias)
        self.ls2 = _LayerScale(dim, init_values)


class DinoVisionTransformer(nn.Module):
    """models/conditioner/dinov2/models/vision_transformer.py:DinoVisionTransformer, inference only."""

    LN_EPS = 1e-6

    def __init__(self, img_size=518, patch_size=14, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, ffn_bias=True, proj_bias=True, init_values=1.0, ffn_layer="mlp", block_chunks=0,
                 num_register_tokens=4, interpolate_antialias=True, interpolate_offset=0.0, modulation_dim=None):
        super().__init__()
        if modulation_dim is not None or ffn_layer != "mlp" or block_chunks != 0 or not init_values:
            raise NotImplementedError("only the shipped dinov2_vit*14_reg configuration (plain Mlp blocks with "
                                      "LayerScale, no modulation, no block chunks) is implemented")
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        self.n_blocks = depth
        self.num_heads = num_heads
        self.patch_size = patch_size
        self.num_register_tokens = num_register_tokens
        self.interpolate_antialias = interpolate_antialias
        self.interpolate_offset = interpolate_offset
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        self.register_tokens = nn.Parameter(torch.zeros(1, num_register_tokens, embed_dim)) if num_register_tokens else None
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias, init_values)
                                     for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.__dict__["_pack"] = {}
        self.__dict__["_pos_cache"] = {}

    # ------------------------------------------------------------------ packed 16-bit weights
    def packed(self, dtype: torch.dtype) -> Dict:
        ver = tuple(p._version for p in self.parameters())
        key = (dtype, str(self.cls_token.device), ver)
        if key in self._pack:
            return self._pack[key]
        with torch.no_grad():
            def ln(norm):   # LayerNorm affine as the (1 + scale, shift) modulation of primx_layernorm_modulate
                return _c16(norm.weight - 1.0, dtype).view(1, -1), _c16(norm.bias, dtype).view(1, -1)
            blocks = []
            for b in self.blocks:
                n1s, n1b = ln(b.norm1)
                n2s, n2b = ln(b.norm2)
                blocks.append({
                    "n1_scale": n1s, "n1_shift": n1b, "n2_scale": n2s, "n2_shift": n2b,
                    "w_qkv": _c16(b.attn.qkv.weight, dtype), "b_qkv": _c16(b.attn.qkv.bias, dtype) if b.attn.qkv.bias is not None else None,
                    "w_proj": _c16(b.attn.proj.weight, dtype), "b_proj": _c16(b.attn.proj.bias, dtype) if b.attn.proj.bias is not None else None,
                    "w_fc1": _c16(b.mlp.fc1.weight, dtype), "b_fc1": _c16(b.mlp.fc1.bias, dtype) if b.mlp.fc1.bias is not None else None,
                    "w_fc2": _c16(b.mlp.fc2.weight, dtype), "b_fc2": _c16(b.mlp.fc2.bias, dtype) if b.mlp.fc2.bias is not None else None,
                    "ls1": _c16(b.ls1.gamma, dtype).view(1, -1), "ls2": _c16(b.ls2.gamma, dtype).view(1, -1),
                })
            ns, nb = ln(self.norm)
            pk = {"blocks": blocks, "n_scale": ns, "n_shift": nb,
                  "w_patch": self.patch_embed.proj.weight.detach().reshape(self.embed_dim, -1).float().contiguous(),
                  "b_patch": self.patch_embed.proj.bias.detach().float().contiguous()}
        self.__dict__["_pack"] = {key: pk}
        return pk

    # ------------------------------------------------------------------ positional table
    def interpolate_pos_encoding(self, npatch: int, w: int, h: int) -> torch.Tensor:
        """[1 + npatch, D] fp32 (vision_transformer.py:188-216).  At the training resolution (518 for the shipped
        model) this is `pos_embed` itself; other sizes resample the patch grid ONCE per size with the reference's own
        call (bicubic `F.interpolate`, a constant-table precompute that is cached)."""
        pe = self.pos_embed.detach().float()
        N = pe.shape[1] - 1
        if npatch == N and w == h:
            return pe[0].contiguous()
        key = (w, h, pe._version, str(pe.device))
        tab = self._pos_cache.get(key)
        if tab is None:
            dim = pe.shape[-1]
            w0 = w // self.patch_size + self.interpolate_offset
            h0 = h // self.patch_size + self.interpolate_offset
            sqrt_n = math.sqrt(N)
            grid = nn.functional.interpolate(
                pe[:, 1:].reshape(1, int(sqrt_n), int(sqrt_n), dim).permute(0, 3, 1, 2),
                scale_factor=(float(w0) / sqrt_n, float(h0) / sqrt_n), mode="bicubic", antialias=self.interpolate_antialias)
            assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
            tab = torch.cat([pe[0, :1], grid.permute(0, 2, 3, 1).reshape(-1, dim)], dim=0).contiguous()
            self.__dict__["_pos_cache"] = {key: tab}
        return tab

    # ------------------------------------------------------------------ forward
    @ops.on_input_device
    def forward_features(self, x: torch.Tensor, masks=None, mod=None, precision_dtype: torch.dtype = torch.float16):
        """x: [B, 3, H, W] preprocessed image (fp32) -> the reference's dict (vision_transformer.py:266-283)."""
        if masks is not None or mod is not None:
            raise NotImplementedError("masking / modulation are not part of the shipped conditioner")
        if self.training:
            raise NotImplementedError("the accelerated DINOv2 is inference-only: call .eval()")
        if not x.is_cuda:
            raise RuntimeError("DinoVisionTransformer.forward_features needs HIP device tensors; there is no CPU path")
        if precision_dtype not in (torch.float16, torch.bfloat16):
            raise NotImplementedError("16-bit GEMM operands only (precision_dtype fp16 / bf16)")
        dt = precision_dtype
        B, C, Hh, Ww = x.shape
        ps = self.patch_size
        if Hh % ps or Ww % ps:
            raise AssertionError(f"image size {Hh}x{Ww} is not a multiple of the patch size {ps}")
        hp, wp = Hh // ps, Ww // ps
        npatch = hp * wp
        D, H = self.embed_dim, self.num_heads
        dh = D // H
        pk = self.packed(dt)

        # patch embedding = non-overlapping conv = GEMM over unfolded patches (layout change only here), fp32
        cols = x.float().reshape(B, C, hp, ps, wp, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * npatch, C * ps * ps).contiguous()
        patches = ops.linear_f32(cols, pk["w_patch"], pk["b_patch"]).view(B, npatch, D)
        pos = self.interpolate_pos_encoding(npatch, Hh, Ww)
        reg = self.register_tokens.detach().float()[0].contiguous() if self.register_tokens is not None else None
        h = ops.vit_tokens(patches, self.cls_token.detach().float().reshape(D).contiguous(), pos, reg)   # [B, nt, D] fp32
        nt = h.shape[1]
        T = B * nt
        h = h.view(T, D)
        xn = torch.empty(T, D, dtype=dt, device=x.device)
        hid = torch.empty(T, pk["blocks"][0]["w_fc1"].shape[0], dtype=dt, device=x.device) if self.n_blocks else None
        for w in pk["blocks"]:
            # x = x + ls1(attn(norm1(x)))      (layers/block.py:90-115, layers/attention.py:56-89)
            ops.layernorm_modulate(h, w["n1_shift"], w["n1_scale"], T, xn, self.LN_EPS)
            qkv = ops.linear(xn, w["w_qkv"], w["b_qkv"]).view(B, nt, 3, H, dh)
            att = ops.memory_efficient_attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])        # scale dh^-0.5
            ops.linear_gate_residual(att.reshape(T, D), w["w_proj"], w["b_proj"], w["ls1"], h, T)
            # x = x + ls2(mlp(norm2(x)))       (layers/mlp.py: fc1 -> GELU (exact) -> fc2)
            ops.layernorm_modulate(h, w["n2_shift"], w["n2_scale"], T, xn, self.LN_EPS)
            ops.linear(xn, w["w_fc1"], w["b_fc1"], out=hid, act=ACT_GELU_ERF)
            ops.linear_gate_residual(hid, w["w_fc2"], w["b_fc2"], w["ls2"], h, T)
        ops.layernorm_modulate(h, pk["n_shift"], pk["n_scale"], T, xn, self.LN_EPS)
        x_norm = xn.view(B, nt, D).float()
        R = self.num_register_tokens
        return {"x_norm_clstoken": x_norm[:, 0], "x_norm_regtokens": x_norm[:, 1:R + 1],
                "x_norm_patchtokens": x_norm[:, R + 1:], "x_prenorm": h.view(B, nt, D), "masks": masks}

    def forward(self, x, is_training=False, **kwargs):
        ret = self.forward_features(x, **kwargs)
        if is_training:
            return ret
        raise NotImplementedError("the classification head is Identity in the reference and unused by 3DTopia-XL; "
                                  "call with is_training=True as Dinov2Wrapper does (image_dinov2.py:51)")

    def conditioner_tokens(self, x: torch.Tensor, precision_dtype: torch.dtype = torch.float16) -> torch.Tensor:
        """[B, 1 + n_patches, D]: what Dinov2Wrapper.forward returns for a preprocessed image (image_dinov2.py:56-61)."""
        out = self.forward_features(x, precision_dtype=precision_dtype)
        return torch.cat([out["x_norm_clstoken"].unsqueeze(1), out["x_norm_patchtokens"]], dim=1)


def vit_base(patch_size=14, num_register_tokens=4, **kwargs):
    """dinov2_vitb14_reg (hub/backbones.py: img 518, patch 14, init_values 1.0, 4 registers, antialias, offset 0)."""
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                                 num_register_tokens=num_register_tokens, **kwargs)
