"""PrimX 3D-VAE with the decode half on the HIP path - drop-in for ``models.vae3d_dib.VAE``.

Same constructor kwargs and the same ``state_dict`` keys as the reference (models/vae3d_dib.py:389-453),
including the encoder and ``quant_conv`` tensors that inference never touches (checkpoints are loaded
strictly, inference.py:257-258).  Only ``decode`` computes; ``encode``/``forward`` are training-side
and raise.

Decoder data layout (MI355X-first): activations are channels-LAST 16-bit ``[P, V, C]`` (P primitives,
V = S^3 voxels in z,y,x raster order), so every 3x3x3 convolution is an implicit GEMM on MFMA whose A
rows are gathered neighbour voxels with a contiguous channel vector (csrc/gemm.hip, GATHER), the
k2s2 transposed convolution is one GEMM with N = 8*Cout and a scatter epilogue, the 1x1 shortcut is a
plain GEMM, and the skip connection ``(x + shortcut(res)) * sqrt(.5)`` is fused into conv2's epilogue.
GroupNorm statistics stay fp32.  The reference runs this decoder in fp32 (TF32 convolutions under
the CLI's allow_tf32 flags, inference.py:377-380); here the storage/MFMA-input type is
``compute_dtype`` (fp16 by default: 10 mantissa bits, the same input precision as TF32) with fp32
accumulation - the tolerance is stated in tests/test_hip_vae.py and DESIGN.md.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from ._lib import HEADS_KROWS, HEADS_ROWS, HEADS_VT
from .attention import MemEffAttention


class VolumeAttention(nn.Module):
    """GroupNorm -> MemEffAttention over the voxels -> (x + res) * skip_scale (vae3d_dib.py:12-48)."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, proj_bias: bool = True,
                 attn_drop: float = 0.0, proj_drop: float = 0.0, groups: int = 32, eps: float = 1e-5,
                 residual: bool = True, skip_scale: float = 1):
        super().__init__()
        self.residual = residual
        self.skip_scale = skip_scale
        self.norm = nn.GroupNorm(num_groups=groups, num_channels=dim, eps=eps, affine=True)
        self.attn = MemEffAttention(dim, num_heads, qkv_bias, proj_bias, attn_drop, proj_drop)


class ResnetBlock(nn.Module):
    """GN-SiLU-conv3 x2 + skip (vae3d_dib.py:93-145); only resample='default' exists on this path."""

    def __init__(self, in_channels: int, out_channels: int, resample: str = "default", groups: int = 32,
                 eps: float = 1e-5, skip_scale: float = 1):
        super().__init__()
        if resample != "default":
            raise NotImplementedError("resampling ResnetBlocks are not used by the shipped VAE")
        self.in_channels, self.out_channels, self.skip_scale = in_channels, out_channels, skip_scale
        self.norm1 = nn.GroupNorm(num_groups=min(groups, in_channels), num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv3d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = nn.GroupNorm(num_groups=min(groups, out_channels), num_channels=out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv3d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.shortcut = nn.Identity()
        if in_channels != out_channels:
            self.shortcut = nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=True)


class DownBlock(nn.Module):
    """Encoder stage - parameters only (vae3d_dib.py:147-184)."""

    def __init__(self, in_channels, out_channels, num_layers=1, downsample=True, skip_scale=1,
                 gradient_checkpointing=False):
        super().__init__()
        self.nets = nn.ModuleList([ResnetBlock(in_channels if i == 0 else out_channels, out_channels,
                                               skip_scale=skip_scale) for i in range(num_layers)])
        self.downsample = nn.Conv3d(out_channels, out_channels, kernel_size=3, stride=2, padding=1) if downsample else None


class MidBlock(nn.Module):
    """ResnetBlock, then (attention, ResnetBlock) x num_layers (vae3d_dib.py:187-226)."""

    def __init__(self, in_channels, num_layers=1, attention=True, attention_heads=8, skip_scale=1,
                 gradient_checkpointing=False):
        super().__init__()
        nets = [ResnetBlock(in_channels, in_channels, skip_scale=skip_scale)]
        attns = []
        for _ in range(num_layers):
            nets.append(ResnetBlock(in_channels, in_channels, skip_scale=skip_scale))
            attns.append(VolumeAttention(in_channels, attention_heads, skip_scale=skip_scale) if attention else None)
        self.nets = nn.ModuleList(nets)
        self.attns = nn.ModuleList(attns)


class UpBlock(nn.Module):
    """ResnetBlocks then optional ConvTranspose3d(k2, s2) (vae3d_dib.py:229-267)."""

    def __init__(self, in_channels, out_channels, num_layers=1, upsample=True, skip_scale=1,
                 gradient_checkpointing=False):
        super().__init__()
        self.nets = nn.ModuleList([ResnetBlock(in_channels if i == 0 else out_channels, out_channels,
                                               skip_scale=skip_scale) for i in range(num_layers)])
        self.upsample = nn.ConvTranspose3d(out_channels, out_channels, kernel_size=2, stride=2) if upsample else None


class Encoder(nn.Module):
    """Training-side half: holds the checkpoint's tensors, never runs (vae3d_dib.py:270-327)."""

    def __init__(self, in_channels=1, out_channels=32, down_channels=(8, 16, 32, 64), mid_attention=True,
                 layers_per_block=2, skip_scale=np.sqrt(0.5), gradient_checkpointing=False):
        super().__init__()
        self.conv_in = nn.Conv3d(in_channels, down_channels[0], kernel_size=3, stride=1, padding=1)
        blocks, cout = [], down_channels[0]
        for i, ch in enumerate(down_channels):
            cin, cout = cout, ch
            blocks.append(DownBlock(cin, cout, num_layers=layers_per_block, downsample=(i != len(down_channels) - 1),
                                    skip_scale=skip_scale))
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock(down_channels[-1], attention=mid_attention, skip_scale=skip_scale)
        self.norm_out = nn.GroupNorm(num_channels=down_channels[-1], num_groups=32, eps=1e-5)
        self.conv_out = nn.Conv3d(down_channels[-1], out_channels, kernel_size=3, stride=1, padding=1)


class Decoder(nn.Module):
    """conv_in -> MidBlock -> UpBlocks -> GN/SiLU/conv_out (vae3d_dib.py:330-387)."""

    def __init__(self, in_channels=16, out_channels=1, up_channels=(64, 32, 16, 8), mid_attention=True,
                 layers_per_block=2, skip_scale=np.sqrt(0.5), gradient_checkpointing=False):
        super().__init__()
        self.conv_in = nn.Conv3d(in_channels, up_channels[0], kernel_size=3, stride=1, padding=1)
        self.mid_block = MidBlock(up_channels[0], attention=mid_attention, skip_scale=skip_scale)
        blocks, cout = [], up_channels[0]
        for i, ch in enumerate(up_channels):
            cin, cout = cout, ch
            blocks.append(UpBlock(cin, cout, num_layers=layers_per_block, upsample=(i != len(up_channels) - 1),
                                  skip_scale=skip_scale))
        self.up_blocks = nn.ModuleList(blocks)
        self.norm_out = nn.GroupNorm(num_channels=up_channels[-1], num_groups=min(32, up_channels[-1]), eps=1e-5)
        self.conv_out = nn.ConvTranspose3d(up_channels[-1], out_channels, kernel_size=3, stride=1, padding=1)


def _conv_weight_as_gemm(w: torch.Tensor, dtype) -> torch.Tensor:
    """[Cout, Cin, 3,3,3] -> [Cout, Kpad], k = tap*Cin + ci, tap = (dz*3+dy)*3+dx, zero-padded to a multiple of 64."""
    cout, cin = w.shape[0], w.shape[1]
    k = 27 * cin
    kpad = (k + 63) // 64 * 64
    out = torch.zeros(cout, kpad, dtype=dtype, device=w.device)
    out[:, :k] = w.permute(0, 2, 3, 4, 1).reshape(cout, k).to(dtype)
    return out


class VAE(nn.Module):
    """models/vae3d_dib.py:389-453 with an accelerated ``decode``."""

    def __init__(self, in_channels: int = 1, latent_channels: int = 16, out_channels: int = 1,
                 down_channels: Sequence[int] = (16, 32, 64, 128, 256), mid_attention: bool = True,
                 up_channels: Sequence[int] = (256, 128, 64, 32, 16), layers_per_block: int = 2,
                 skip_scale: float = np.sqrt(0.5), gradient_checkpointing: bool = False):
        super().__init__()
        self.latent_channels = latent_channels
        self.skip_scale = float(skip_scale)
        self.encoder = Encoder(in_channels=in_channels, out_channels=2 * latent_channels, down_channels=down_channels,
                               mid_attention=mid_attention, layers_per_block=layers_per_block, skip_scale=skip_scale)
        self.decoder = Decoder(in_channels=latent_channels, out_channels=out_channels, up_channels=up_channels,
                               mid_attention=mid_attention, layers_per_block=layers_per_block, skip_scale=skip_scale)
        self.quant_conv = nn.Conv3d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv3d(latent_channels, latent_channels, 1)
        self.compute_dtype = torch.float16
        self._pack: Dict = {}

    # ------------------------------------------------------------------ packing
    def repack(self) -> None:
        self._pack = {}

    def _apply(self, fn, *a, **k):
        self.__dict__["_pack"] = {}
        self.__dict__["_attn_ws"] = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.repack()
        return super().load_state_dict(*a, **k)

    def _res_pack(self, blk: ResnetBlock, dt) -> Dict:
        c16 = lambda t: t.detach().to(dt).contiguous()
        f32 = lambda t: t.detach().float().contiguous()
        d = {"g1": f32(blk.norm1.weight), "b1": f32(blk.norm1.bias), "groups1": blk.norm1.num_groups,
             "eps1": blk.norm1.eps, "w1": _conv_weight_as_gemm(blk.conv1.weight.detach(), dt), "c1": c16(blk.conv1.bias),
             "g2": f32(blk.norm2.weight), "b2": f32(blk.norm2.bias), "groups2": blk.norm2.num_groups,
             "eps2": blk.norm2.eps, "w2": _conv_weight_as_gemm(blk.conv2.weight.detach(), dt), "c2": c16(blk.conv2.bias),
             "wsc": None, "csc": None}
        # weight images of the activation-resident kernels (None for other shapes; each is used only on the grid it is for)
        if isinstance(blk.shortcut, nn.Conv3d):
            d["wsc"] = c16(blk.shortcut.weight.reshape(blk.out_channels, blk.in_channels))
            d["csc"] = c16(blk.shortcut.bias)
        # (the 256 -> 32 block's image carries its 1x1 shortcut as a 28th weight block: ops.conv3d_s8_fused)
        d["w1p"] = ops.pack_conv3(d["w1"], blk.conv1.in_channels,
                                  Wsc=d["wsc"] if (blk.in_channels, blk.out_channels) == (256, 32) else None)
        d["w2p"] = ops.pack_conv3(d["w2"], blk.conv2.in_channels)
        return d

    def packed(self, dt: torch.dtype) -> Dict:
        key = (dt, self.post_quant_conv.weight.device)
        if key in self._pack:
            return self._pack[key]
        if self.latent_channels != 1:
            raise NotImplementedError("the accelerated decoder covers latent_channels == 1 (the shipped PrimX VAE)")
        dec = self.decoder
        c16 = lambda t: t.detach().to(dt).contiguous()
        f32 = lambda t: t.detach().float().contiguous()
        with torch.no_grad():
            pk = {
                # post_quant_conv is a 1x1x1 conv on ONE channel = scalar affine (vae3d_dib.py:429,438)
                "pq_a": float(self.post_quant_conv.weight.reshape(-1)[0]), "pq_b": float(self.post_quant_conv.bias[0]),
                "w_in": f32(dec.conv_in.weight.reshape(dec.conv_in.out_channels, 27)), "b_in": f32(dec.conv_in.bias),
                "mid": [self._res_pack(n, dt) for n in dec.mid_block.nets],
                "attn": [],
                "up": [],
                "g_out": f32(dec.norm_out.weight), "b_out": f32(dec.norm_out.bias), "groups_out": dec.norm_out.num_groups,
                "eps_out": dec.norm_out.eps,
                # ConvTranspose3d(k3, s1, p1) == Conv3d with the kernel flipped and in/out swapped (vae3d_dib.py:367)
                "w_out": _conv_weight_as_gemm(dec.conv_out.weight.detach().flip(2, 3, 4).permute(1, 0, 2, 3, 4), dt),
                "c_out": c16(dec.conv_out.bias),
            }
            pk["w_outp"] = ops.pack_conv3(pk["w_out"], dec.conv_out.in_channels)
            for a in dec.mid_block.attns:
                if a is None:
                    pk["attn"].append(None)
                    continue
                pk["attn"].append({
                    "g": f32(a.norm.weight), "b": f32(a.norm.bias), "groups": a.norm.num_groups, "eps": a.norm.eps,
                    "w_qkv": c16(a.attn.qkv.weight), "b_qkv": None if a.attn.qkv.bias is None else c16(a.attn.qkv.bias),
                    "w_proj": c16(a.attn.proj.weight),
                    "b_proj": None if a.attn.proj.bias is None else c16(a.attn.proj.bias),
                    "heads": a.attn.num_heads, "residual": a.residual,
                })
            for ub in dec.up_blocks:
                u = {"nets": [self._res_pack(n, dt) for n in ub.nets], "w_up": None, "c_up": None, "w_upp": None}
                if ub.upsample is not None:
                    w = ub.upsample.weight.detach()  # [Cin, Cout, 2, 2, 2]
                    u["w_up"] = c16(w.permute(2, 3, 4, 1, 0).reshape(8 * w.shape[1], w.shape[0]))
                    u["c_up"] = c16(ub.upsample.bias)
                    u["w_upp"] = ops.pack_convt_s4(u["w_up"])
                pk["up"].append(u)
        self._pack = {key: pk}
        return pk

    # ------------------------------------------------------------------ decode
    def _resnet(self, h: torch.Tensor, w: Dict, S: int, gn_part=None) -> torch.Tensor:
        """gn_part = (partial GroupNorm sums of h, their shifts) when h comes straight from the upsample kernel."""
        P, V, Cin = h.shape
        if gn_part is not None and self._fusable_front(w, S):
            # norm1 + SiLU + conv1 and the 1x1 shortcut in one kernel, on the raw upsample output
            t, res = ops.conv3d_s8_fused(h, w["w1p"], w["c1"], gn_part[0], gn_part[1], w["g1"], w["b1"], w["eps1"], w["csc"])
            if ops.conv3_takes_groupnorm(w["w2p"], S, w["groups2"]):
                return ops.conv3d_k3(t, w["w2"], w["c2"], S, res=res, res_scale=self.skip_scale, Wp=w["w2p"],
                                     gn=(w["g2"], w["b2"], w["eps2"]))
            t = ops.groupnorm_silu(t, w["g2"], w["b2"], w["groups2"], w["eps2"], True)
            return ops.conv3d_k3(t, w["w2"], w["c2"], S, res=res, res_scale=self.skip_scale, Wp=w["w2p"])
        # GroupNorm + SiLU goes into the convolution kernel where that kernel holds the whole primitive (8^3 x 32 channels)
        if ops.conv3_takes_groupnorm(w["w1p"], S, w["groups1"]):
            t = ops.conv3d_k3(h, w["w1"], w["c1"], S, Wp=w["w1p"], gn=(w["g1"], w["b1"], w["eps1"]))
        else:
            t = ops.groupnorm_silu(h, w["g1"], w["b1"], w["groups1"], w["eps1"], True)
            t = ops.conv3d_k3(t, w["w1"], w["c1"], S, Wp=w["w1p"])
        res = h
        if w["wsc"] is not None:
            res = ops.linear_residual(h.view(P * V, Cin), w["wsc"], w["csc"], None, 1.0).view(P, V, -1)
        if ops.conv3_takes_groupnorm(w["w2p"], S, w["groups2"]):
            return ops.conv3d_k3(t, w["w2"], w["c2"], S, res=res, res_scale=self.skip_scale, Wp=w["w2p"],
                                 gn=(w["g2"], w["b2"], w["eps2"]))
        t = ops.groupnorm_silu(t, w["g2"], w["b2"], w["groups2"], w["eps2"], True)
        return ops.conv3d_k3(t, w["w2"], w["c2"], S, res=res, res_scale=self.skip_scale, Wp=w["w2p"])

    @staticmethod
    def _fusable_front(w: Dict, S: int) -> bool:
        p = w.get("w1p")
        return p is not None and p.kind == "s8" and p.has_sc and p.S == S and w["groups1"] == 32 and w["wsc"] is not None

    def _attention(self, h: torch.Tensor, w: Dict) -> torch.Tensor:
        P, V, Cc = h.shape
        H = w["heads"]
        dh = Cc // H
        t = ops.groupnorm_silu(h, w["g"], w["b"], w["groups"], w["eps"], False)
        # persistent zero-padded operand buffers (the kernels never write the pads, so they stay zero): three 67 MB zero fills
        # per decode otherwise.  One entry per (P, dtype, device); bounded.
        # 64 voxels x 32-wide heads (the shipped decoder): compact operands for the one-wave-per-problem kernel
        # (csrc/attention.hip attn64_kernel); anything else: the 256-row workgroup kernel's padding
        pad = 64 if (V <= 64 and dh == 32) else ops.BQ

        def alloc():
            return (ops.alloc_heads(P, H, V, dh, HEADS_ROWS, h.dtype, h.device, pad, "q"),
                    ops.alloc_heads(P, H, V, dh, HEADS_KROWS, h.dtype, h.device, pad, "k"),
                    ops.alloc_heads(P, H, V, dh, HEADS_VT, h.dtype, h.device, pad))
        if P <= 4096:                                      # (200 MB at P = 2048; larger one-off batches are not kept)
            key = (P, H, V, dh, pad, h.dtype, h.device)
            ws = self.__dict__.setdefault("_attn_ws", {})
            if key not in ws:
                if len(ws) >= 2:
                    ws.clear()
                ws[key] = alloc()
            Q, K, Vt = ws[key]
        else:
            Q, K, Vt = alloc()
        ops.linear_heads(t.view(P * V, Cc), w["w_qkv"], w["b_qkv"], V, H, dh, [HEADS_ROWS, HEADS_KROWS, HEADS_VT],
                         [Q, K, Vt], Q.shape[2])
        att = ops.attention(Q, K, Vt, V, V, dh, dh ** -0.5)
        if w["residual"]:
            out = ops.linear_residual(att.view(P * V, Cc), w["w_proj"], w["b_proj"], h.view(P * V, Cc), self.skip_scale)
        else:
            out = ops.linear_residual(att.view(P * V, Cc), w["w_proj"], w["b_proj"], None, 1.0)
        return out.view(P, V, Cc)

    @ops.on_input_device
    def decode(self, x: torch.Tensor, denormalize: bool = False) -> torch.Tensor:
        """x: (P, 1, S, S, S) fp32 latents -> (P, out_channels, 2S, 2S, 2S) fp32  (vae3d_dib.py:437-440).

        ``denormalize=True`` additionally applies the caller's inverse normalisation
        (inference.py:345-346: channel 0 / 5, others (v + 1) / 2) inside the output kernel."""
        if not x.is_cuda:
            raise RuntimeError("VAE.decode needs HIP device tensors; there is no CPU path")
        P, Cl, S = x.shape[0], x.shape[1], x.shape[2]
        if Cl != self.latent_channels or x.shape[3] != S or x.shape[4] != S:
            raise AssertionError("latent must be (P, latent_channels, S, S, S)")
        dt = self.compute_dtype
        pk = self.packed(dt)
        with torch.no_grad():
            z = x.reshape(P, S * S * S).float().contiguous()
            h = ops.conv_in(z, pk["pq_a"], pk["pq_b"], pk["w_in"], pk["b_in"], S, dt)
            h = self._resnet(h, pk["mid"][0], S)
            for aw, rw in zip(pk["attn"], pk["mid"][1:]):
                if aw is not None:
                    h = self._attention(h, aw)
                h = self._resnet(h, rw, S)
            gn_part = None
            for iu, u in enumerate(pk["up"]):
                for k, rw in enumerate(u["nets"]):
                    h = self._resnet(h, rw, S, gn_part if k == 0 else None)
                gn_part = None
                if u["w_up"] is not None:
                    # the upsample kernel also leaves the partial GroupNorm sums of its output when the next block can take them
                    nxt = pk["up"][iu + 1]["nets"][0] if iu + 1 < len(pk["up"]) else None
                    want = u["w_upp"] is not None and S == 4 and nxt is not None and self._fusable_front(nxt, 2 * S)
                    if want:
                        h, part = ops.convtranspose_k2s2(h, u["w_up"], u["c_up"], S, Wp=u["w_upp"], want_stats=True)
                        gn_part = (part, u["c_up"])
                    else:
                        h = ops.convtranspose_k2s2(h, u["w_up"], u["c_up"], S, Wp=u["w_upp"])
                    S *= 2
            if ops.conv3_takes_groupnorm(pk["w_outp"], S, pk["groups_out"]):
                h = ops.conv3d_k3(h, pk["w_out"], pk["c_out"], S, Wp=pk["w_outp"], gn=(pk["g_out"], pk["b_out"], pk["eps_out"]))
            else:
                h = ops.groupnorm_silu(h, pk["g_out"], pk["b_out"], pk["groups_out"], pk["eps_out"], True)
                h = ops.conv3d_k3(h, pk["w_out"], pk["c_out"], S)
            out = ops.vae_output(h, denormalize)
        return out.view(P, -1, S, S, S)

    def encode(self, x):
        raise NotImplementedError("the encoder is training-only and outside the accelerated path")

    def forward(self, x, sample=True):
        raise NotImplementedError("VAE.forward (encode + decode) is training-only; use decode()")
