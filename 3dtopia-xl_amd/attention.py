"""Attention modules with the reference's names, constructor arguments and ``state_dict`` keys
(models/attention.py:20-59 ``MemEffAttention``, :62-114 ``MemEffCrossAttention``), computing on the
HIP path: projection GEMMs whose epilogues write the attention operand layouts directly, then
``primx_attention``.  No xFormers.

The modules are parameter containers for the fused DiT / VAE drivers (which read their packed
16-bit weights) and can also be called on their own; standalone calls compute in ``compute_dtype``
(fp16 unless changed) with fp32 accumulation and return a tensor of that dtype.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import ops
from ._lib import HEADS_KROWS, HEADS_ROWS, HEADS_VT


class _PackedMixin:
    """Caches 16-bit copies of the module's weights; dropped whenever parameters are re-assigned."""

    def _packed(self, dtype: torch.dtype) -> Dict[str, Optional[torch.Tensor]]:
        cache = self.__dict__.setdefault("_pack_cache", {})
        key = (dtype, next(self.parameters()).device)
        if key not in cache:
            cache.clear()
            cache[key] = self._build_pack(dtype)
        return cache[key]

    def repack(self) -> None:
        self.__dict__.get("_pack_cache", {}).clear()

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / .float() ...
        self.repack()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):  # reached for every sub-module of a load_state_dict()
        self.repack()
        return super()._load_from_state_dict(*a, **k)


def _c16(p: Optional[torch.Tensor], dtype) -> Optional[torch.Tensor]:
    return None if p is None else p.detach().to(dtype).contiguous()


class MemEffAttention(_PackedMixin, nn.Module):
    """Fused-QKV self-attention (models/attention.py:20-59).  Logits scale ``head_dim**-0.5``."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, proj_bias: bool = True,
                 attn_drop: float = 0.0, proj_drop: float = 0.0, gradient_checkpointing: bool = False) -> None:
        super().__init__()
        if attn_drop or proj_drop:
            raise NotImplementedError("dropout is training-only; the accelerated path is inference")
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.gradient_checkpointing = gradient_checkpointing
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self.compute_dtype = torch.float16

    def _build_pack(self, dtype):
        return {"w_qkv": _c16(self.qkv.weight, dtype), "b_qkv": _c16(self.qkv.bias, dtype),
                "w_proj": _c16(self.proj.weight, dtype), "b_proj": _c16(self.proj.bias, dtype)}

    @ops.on_input_device
    def forward(self, x: torch.Tensor, attn_bias=None) -> torch.Tensor:
        if attn_bias is not None:
            raise NotImplementedError("attn_bias is not used on the 3DTopia-XL path")
        B, N, Cc = x.shape
        dt = x.dtype if x.dtype in (torch.float16, torch.bfloat16) else self.compute_dtype
        a = x.reshape(B * N, Cc)
        a = ops.cast16(a.contiguous(), dt) if a.dtype == torch.float32 else a.contiguous()
        w = self._packed(dt)
        n_pad = ops.round_up(N, ops.BQ)
        H, dh = self.num_heads, self.head_dim
        Q = ops.alloc_heads(B, H, N, dh, HEADS_ROWS, dt, x.device, ops.BQ, "q")
        K = ops.alloc_heads(B, H, N, dh, HEADS_KROWS, dt, x.device, ops.BQ, "k")
        Vt = ops.alloc_heads(B, H, N, dh, HEADS_VT, dt, x.device, ops.BQ)
        ops.linear_heads(a, w["w_qkv"], w["b_qkv"], N, H, dh, [HEADS_ROWS, HEADS_KROWS, HEADS_VT], [Q, K, Vt], n_pad)
        att = ops.attention(Q, K, Vt, N, N, dh, self.scale)
        out = ops.linear(att.view(B * N, Cc), w["w_proj"], w["b_proj"])
        return out.view(B, N, Cc)


class MemEffCrossAttention(_PackedMixin, nn.Module):
    """Cross-attention (models/attention.py:62-114).  ``q`` is pre-multiplied by ``scale`` AND the
    attention core applies ``head_dim**-0.5`` again (attention.py:105,109): logits = q.k / head_dim."""

    def __init__(self, dim: int, dim_q: int, dim_k: int, dim_v: int, num_heads: int = 8, qkv_bias: bool = False,
                 proj_bias: bool = True, attn_drop: float = 0.0, proj_drop: float = 0.0,
                 gradient_checkpointing: bool = False) -> None:
        super().__init__()
        if attn_drop or proj_drop:
            raise NotImplementedError("dropout is training-only; the accelerated path is inference")
        if dim_k != dim_v:
            raise NotImplementedError("k and v share the conditioning tensor on this path (dim_k == dim_v)")
        self.dim = dim
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.gradient_checkpointing = gradient_checkpointing
        self.to_q = nn.Linear(dim_q, dim, bias=qkv_bias)
        self.to_k = nn.Linear(dim_k, dim, bias=qkv_bias)
        self.to_v = nn.Linear(dim_v, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self.compute_dtype = torch.float16

    def _build_pack(self, dtype):
        has_b = self.to_k.bias is not None
        return {
            "w_q": _c16(self.to_q.weight, dtype), "b_q": _c16(self.to_q.bias, dtype),
            "w_kv": _c16(torch.cat([self.to_k.weight, self.to_v.weight], 0), dtype),
            "b_kv": _c16(torch.cat([self.to_k.bias, self.to_v.bias], 0), dtype) if has_b else None,
            "w_proj": _c16(self.proj.weight, dtype), "b_proj": _c16(self.proj.bias, dtype),
        }

    @ops.on_input_device
    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_bias=None) -> torch.Tensor:
        if attn_bias is not None:
            raise NotImplementedError("attn_bias is not used on the 3DTopia-XL path")
        if k is not v and not (k.data_ptr() == v.data_ptr() and k.shape == v.shape):
            raise NotImplementedError("k and v must be the same conditioning tensor")
        B, N, _ = q.shape
        M = k.shape[1]
        dt = q.dtype if q.dtype in (torch.float16, torch.bfloat16) else self.compute_dtype
        to16 = lambda t: ops.cast16(t.contiguous(), dt) if t.dtype == torch.float32 else t.contiguous()
        a = to16(q.reshape(B * N, -1))
        c = to16(k.reshape(B * M, -1))
        w = self._packed(dt)
        H, dh = self.num_heads, self.head_dim
        Q = ops.alloc_heads(B, H, N, dh, HEADS_ROWS, dt, q.device, ops.BQ, "q")
        K = ops.alloc_heads(B, H, M, dh, HEADS_KROWS, dt, q.device, ops.BKV, "k")
        Vt = ops.alloc_heads(B, H, M, dh, HEADS_VT, dt, q.device, ops.BKV)
        ops.linear_heads(a, w["w_q"], w["b_q"], N, H, dh, [HEADS_ROWS], [Q], Q.shape[2], scale0=self.scale)
        ops.linear_heads(c, w["w_kv"], w["b_kv"], M, H, dh, [HEADS_KROWS, HEADS_VT], [K, Vt], K.shape[2])
        att = ops.attention(Q, K, Vt, N, M, dh, self.scale)
        out = ops.linear(att.view(B * N, self.dim), w["w_proj"], w["b_proj"])
        return out.view(B, N, self.dim)
