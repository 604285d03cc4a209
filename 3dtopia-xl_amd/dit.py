"""PrimX diffusion transformer on the HIP path - drop-in for ``models.dit_crossattn.DiT``.

Same constructor kwargs, same ``state_dict`` keys (515 tensors for the shipped config), same
``forward`` / ``forward_with_cfg`` signatures and return shapes as the reference
(models/dit_crossattn.py:111-213).  The compute is re-designed for MI355X:

* the fp32 residual stream ``h`` [B*N, D] stays resident; every projection that adds into it is ONE
  GEMM whose epilogue applies bias, 16-bit rounding, the adaLN gate and the residual add in place;
* LayerNorm + modulate + cast is one row kernel producing the 16-bit GEMM operand;
* q/k/v projections write the attention operand layouts (padded head-major Q/K, transposed
  quad-permuted V) straight from the GEMM epilogue - there is no reshape/unbind/permute pass;
* the adaLN Linear of all blocks + final layer is ONE weight-streaming GEMM per forward
  ([B, D] x [D, depth*9D + 2D]) instead of depth+1 GEMV-shaped launches, and the cross-attention
  to_k / to_v projections of all blocks (same conditioning tokens) are ONE GEMM with N = depth*2D
  writing a [depth, ...] K / V^T cache - fixed per-launch cost dominates these small-K GEMMs;
* rounding points follow the reference's fp16/bf16 autocast topology (fp32 residual stream and
  LayerNorm, 16-bit Linear/attention outputs, 16-bit modulation vectors, 16-bit CFG combine), see
  DESIGN.md section 6.

Two numeric routes, selected exactly as the reference selects them (dit_crossattn.py:184,197):
``enable_amp=True`` with fp16 / bf16 -> the autocast topology on the 16-bit MFMA path (production, everything above);
``enable_amp=False`` (the signature default) or ``precision_dtype=float32`` -> exact fp32 on the fp32 matrix instruction
(csrc/fp32.hip: gfx950 has no TF32, so this is at least the reference's own fp32 / TF32 precision, at 1/16 of the
16-bit MFMA rate).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch
from torch import nn

import ctypes as C

from . import _lib, ops
from ._lib import ACT_GELU_TANH, HEADS_KROWS, HEADS_ROWS, HEADS_VT
from .attention import MemEffAttention, MemEffCrossAttention, _c16


def modulate(x, shift, scale):
    """Reference formula (models/utils.py:19-20); kept for API parity, the HIP path fuses it."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


class TimestepEmbedder(nn.Module):
    """Sinusoid(256) -> Linear -> SiLU -> Linear, fp32 (models/utils.py:27-64)."""

    def __init__(self, hidden_size: int, frequency_embedding_size: int = 256):
        super().__init__()
        self.mlp = nn.Sequential(
            nn.Linear(frequency_embedding_size, hidden_size, bias=True),
            nn.SiLU(),
            nn.Linear(hidden_size, hidden_size, bias=True),
        )
        self.frequency_embedding_size = frequency_embedding_size

    @staticmethod
    def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000) -> torch.Tensor:
        if dim % 2:
            raise NotImplementedError("odd embedding sizes are not used on this path")
        return ops.timestep_embedding(t, dim, float(max_period))

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        f = self.timestep_embedding(t, self.frequency_embedding_size)
        h = ops.linear_f32(f, self.mlp[0].weight.detach(), self.mlp[0].bias.detach(), act_out=1)
        return ops.linear_f32(h, self.mlp[2].weight.detach(), self.mlp[2].bias.detach())


class Mlp(nn.Module):
    """fc1 -> GELU(tanh) -> fc2 parameter container (models/utils.py:66-101; dropouts are p=0)."""

    def __init__(self, in_features: int, hidden_features: Optional[int] = None, out_features: Optional[int] = None,
                 act_layer=None, norm_layer=None, bias: bool = True, drop: float = 0.0, use_conv: bool = False):
        super().__init__()
        if use_conv or norm_layer is not None or drop:
            raise NotImplementedError("only the Linear / no-norm / no-dropout Mlp of the DiT block is supported")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)


class DiTBlock(nn.Module):
    """adaLN-Zero block: cross-attn, self-attn, MLP, each gated (models/dit_crossattn.py:25-58).
    Parameter container; the fused per-block schedule lives in ``DiT._run_block``."""

    def __init__(self, hidden_size, cross_attn_cond_dim, num_heads, mlp_ratio=4.0, proj_bias=False,
                 gradient_checkpointing=False, **block_kwargs):
        super().__init__()
        self.gradient_checkpointing = gradient_checkpointing
        self.crossattn = MemEffCrossAttention(dim=hidden_size, dim_q=hidden_size, dim_k=cross_attn_cond_dim,
                                              dim_v=cross_attn_cond_dim, num_heads=num_heads, qkv_bias=True,
                                              proj_bias=proj_bias, **block_kwargs)
        self.attn = MemEffAttention(dim=hidden_size, num_heads=num_heads, qkv_bias=True, proj_bias=proj_bias,
                                    **block_kwargs)
        self.mlp = Mlp(in_features=hidden_size, hidden_features=int(hidden_size * mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 9 * hidden_size, bias=True))


class FinalLayer(nn.Module):
    """adaLN(2) -> LN -> modulate -> Linear (models/dit_crossattn.py:61-78).  Parameter container."""

    def __init__(self, hidden_size, seq_length, out_channels):
        super().__init__()
        self.linear = nn.Linear(hidden_size, out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))


class DiT(nn.Module):
    """Diffusion transformer over primitive tokens (models/dit_crossattn.py:111-213)."""

    LN_EPS = 1e-6

    def __init__(self, seq_length=2, in_channels=4, condition_channels=512, hidden_size=1152, depth=28,
                 num_heads=16, mlp_ratio=4.0, cond_drop_prob=0.0, attn_proj_bias=False, learn_sigma=True,
                 gradient_checkpointing=False):
        super().__init__()
        self.gradient_checkpointing = gradient_checkpointing
        self.learn_sigma = learn_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.seq_length = seq_length
        self.num_heads = num_heads
        self.hidden_size = hidden_size
        self.depth = depth
        self.condition_channels = condition_channels
        self.cond_drop_prob = cond_drop_prob
        if self.cond_drop_prob > 0:
            self.null_cond_embedding = nn.Parameter(torch.randn(condition_channels))
        self.x_embedder = nn.Linear(in_channels, hidden_size)
        self.t_embedder = TimestepEmbedder(hidden_size)
        self.blocks = nn.ModuleList([
            DiTBlock(hidden_size, condition_channels, num_heads, mlp_ratio=mlp_ratio, proj_bias=attn_proj_bias,
                     gradient_checkpointing=gradient_checkpointing) for _ in range(depth)
        ])
        self.final_layer = FinalLayer(hidden_size, seq_length, self.out_channels)
        self.initialize_weights()
        self._pack: Dict = {}
        self._heads_ws: Dict = {}
        self._heads_owner: Dict = {}   # workspace key -> the set of shape groups that use it (_heads / _heads_begin)
        self._heads_lru: list = []
        self._heads_group = None
        self._cond: Optional[Dict] = None
        self._packed_only = False   # set when only the packed blob was received / loaded, not the fp32 parameters
        # Opt-in exact-algebra shortcut: to_k(y) / to_v(y) do not depend on the timestep
        # (models/attention.py:106-107), so with this flag the K / V projections of all blocks are computed once per
        # conditioning tensor and reused across DDIM steps.  Off by default: every step then projects its conditioning tokens, as
        # the reference does (bench.py quotes its rates against the FLOPs a step executes and reports the algorithmic count of
        # the reference's step next to them).
        self.reuse_cond_kv = False
        # Second opt-in exact-algebra shortcut, CFG forwards only: the unconditional half is conditioned on ONE row repeated L
        # times (`y_null = null_cond_embedding.expand_as(y)`, dit_crossattn.py:207), so its L keys are identical, its softmax
        # is uniform whatever the query, and its cross-attention output is the (identical) value row of every head - to_q and
        # the attention core of that half compute nothing else.  With this flag they are skipped and the value row is broadcast;
        # to_k / to_v, proj and everything else run as before.  Off by default (the headline attends; what it does NOT repeat is
        # the projection of the L identical null rows - `dedup_null_kv` below, 2 % of the algorithmic FLOPs).  Measured at batch 1 (round 3): NO gain (8.55 vs 8.47 ms per step next to reuse_cond_kv) - the
        # half-size to_q and attention launches occupy half of the CUs for the same time; it pays from batch 2 per GPU on.
        self.collapse_null_cross_attention = False
        # Opt-in: run the two classifier-free-guidance halves of `forward_with_cfg` as two concurrent HIP streams
        # (_forward16); identical kernels and results per row.  PRIMX_CFG_STREAMS=1 turns it on for every model.
        self.cfg_streams = os.environ.get("PRIMX_CFG_STREAMS") == "1"
        # forward_with_cfg: the unconditional half's conditioning is ONE embedding expanded to L tokens (dit_crossattn.py:207), so
        # its L rows of to_k / to_v are identical: project 64 + L % 64 of them once per forward and let the cross-attention kernel
        # address them as the L-key sequence (ops.attention(bcast=)) - same tiles, same arithmetic.  PRIMX_NULL_KV_DEDUP=0: the
        # expanded rows are projected, stored and read like the conditional ones.
        self.dedup_null_kv = os.environ.get("PRIMX_NULL_KV_DEDUP", "1") != "0"
        # weight prefetch of the loader-wave GEMMs (_forward16): 2 = carried by the GEMM launches one or two ahead, 1 = carried by the
        # LayerNorm launches, 0 = off (PRIMX_WPREFETCH)
        self.weight_prefetch = int(os.environ.get("PRIMX_WPREFETCH", "2"))
        # the LayerNorm + modulate that follows every gated residual add is requested from the SAME entry point as the GEMM
        # (primx_linear_gate_residual_ln; PRIMX_DIT_FUSE_LN=0: separate primx_layernorm_modulate calls).  `ln_in_kernel`
        # (PRIMX_DIT_LN_TAIL=1) additionally hands the library the sync words that let it run the LayerNorm in the TAIL of the GEMM
        # kernel: bit-identical, and measured SLOWER than the launch it saves in every arrival protocol tried in round 4 (10.25 vs
        # 8.99 ms per step at best, DESIGN.md section 4 / DESIGN_LOG.md section 10) - off by default, kept as a measured experiment.
        self.fuse_ln = os.environ.get("PRIMX_DIT_FUSE_LN", "1") != "0"
        self.ln_in_kernel = os.environ.get("PRIMX_DIT_LN_TAIL", "0") == "1"
        self._ln_sync: Dict = {}              # device -> int32 workspace of the fused route (zero between launches)
        # `fold_ln` (default on; PRIMX_DIT_FOLD=0 turns it off): the LayerNorm fold (include/primx_hip.h, ABI 23; csrc/gemm.hip "LayerNorm fold").  In a PLANNED
        # sampling loop (plan_timesteps: the modulation vectors of every coming call are known) each LayerNorm + modulate between a
        # gated residual add and a Linear is folded into the two GEMMs around it: the gate-residual GEMM also stores the centred,
        # scaled 16-bit operand and partial row sums, the consumer (to_q / qkv / fc1) finishes the statistics and applies them in
        # its epilogue with the per-timestep vectors u = (1 + scale) W^T, v = shift W^T + b.  83 of the 85 LayerNorm launches of a
        # DiT-XL forward disappear (the first of block 0 and the final layer's stay).  NOT bit-identical to the unfolded path: the
        # operand is rounded before the normalisation instead of after it - equal accuracy against the fp32 reference
        # (tools/ln_fold_study.py; tests/test_hip_fold.py).  Applies at the shapes the fold kernels cover (_fold_ok); every other
        # call - unplanned forwards included - takes the LayerNorm launches.
        self.fold_ln = os.environ.get("PRIMX_DIT_FOLD", "1") != "0"
        self._fold_ws: Dict = {}              # (device, rows) -> (center, part) workspaces of the fold
        # the fold's u / v tables are built for a whole planned loop at once (2.5 MB per step for DiT-XL, 3 x depth small GEMMs):
        # loops of more steps than this keep their LayerNorm launches (PRIMX_DIT_FOLD_MAX_STEPS)
        self.fold_max_steps = int(os.environ.get("PRIMX_DIT_FOLD_MAX_STEPS", "128"))
        # `blocks_call` (round 6; PRIMX_DIT_BLOCKS_CALL=0 turns it off): a folded forward hands its block loop to the library in ONE
        # foreign call (primx_dit_blocks_fold, ABI 24: the same entry points with the same arguments, issued from C) instead of 8
        # Python + ctypes calls per block.  Bit-identical; what changes is the host's time per step (tools/host_bound_check.py).
        self.blocks_call = os.environ.get("PRIMX_DIT_BLOCKS_CALL", "1") != "0"
        # `kv_ride` (round 6, ABI 25; PRIMX_DIT_KV_RIDE=0 turns it off): on the one-call route the to_k / to_v projection of the conditioning
        # tokens (attention.py:106-107) is issued by the library - block i + 1's 48 tiles ride on block i's qkv launch, whose 192 tiles leave
        # a quarter of the chip idle at T = 4096 (primx_linear_heads_fold_pair) - instead of one batched launch per forward from here.
        # Same tile kernel, bit-identical operands.
        self.kv_ride = os.environ.get("PRIMX_DIT_KV_RIDE", "1") != "0"
        # Dynamic range (fp16): the folded operand cast16((x - c) rho_p (1 + scale)) is normalised with the PREVIOUS site's (mean, rstd)
        # of the row (ABI 23) - the LayerNorm output up to the factor rho_p / rho by which one gated branch changes the row's spread.
        # Magnitude and spread of the residual stream do not enter (tests/test_hip_fold.py: spreads 3e4 and 1e-5, magnitude 1e4); the
        # reference's own operand cast16(LN (1 + scale) + shift) overflows at |1 + scale| ~ 1900, this one at (rho_p / rho) |1 + scale| ~
        # 1900.  The sampling loop still looks at its final sample once when folded fp16 forwards ran (`fold_overflowed`) and repeats
        # the loop with LayerNorm launches if it is non-finite; bf16 has the range of fp32 and is not checked.
        self._fold_fp16_used = False
        # `block_probe` (diagnostics, tools/validate_checkpoint.py): set to a list and every 16-bit forward appends one record per
        # DiT block - the dynamic range of the fp32 residual stream behind the block and of the 16-bit operand handed to the next
        # Linear (the LayerNorm output, or the fold's operand).  Synchronises after every block: never set on a timed path.
        self.block_probe: Optional[list] = None
        self._side: Dict = {}
        self._t_plan: Optional[Dict] = None   # plan_timesteps(): the coming calls' timesteps and their modulation table

    # ------------------------------------------------------------------ init (dit_crossattn.py:153-182)
    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        for blk in self.blocks:  # adaLN-Zero
            nn.init.zeros_(blk.adaLN_modulation[-1].weight)
            nn.init.zeros_(blk.adaLN_modulation[-1].bias)
        nn.init.zeros_(self.final_layer.adaLN_modulation[-1].weight)
        nn.init.zeros_(self.final_layer.adaLN_modulation[-1].bias)
        nn.init.zeros_(self.final_layer.linear.weight)
        nn.init.zeros_(self.final_layer.linear.bias)

    # ------------------------------------------------------------------ packed 16-bit weights
    def repack(self) -> None:
        self._t_plan = None
        self._pack = {}
        self._heads_ws, self._heads_owner, self._heads_lru, self._heads_group = {}, {}, [], None
        self._cond = None
        self._packed_only = False

    def _apply(self, fn, *a, **k):
        self.__dict__["_pack"] = {}
        self.__dict__["_heads_ws"] = {}
        self.__dict__["_heads_owner"] = {}
        self.__dict__["_heads_lru"] = []
        self.__dict__["_heads_group"] = None
        self.__dict__["_cond"] = None
        self.__dict__["_t_plan"] = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.repack()
        return super().load_state_dict(*a, **k)

    # The packed 16-bit weights live in ONE flat device buffer ("blob": 1.82 GB for DiT-XL) that the per-layer operands
    # are views of: it is filled in place from the fp32 parameters (no concatenation temporaries), it is what a
    # multi-GPU launch broadcasts over xGMI (sharding.broadcast_packed_), and a rank that
    # received it needs no fp32 parameters at all for the 16-bit path.
    _PACK_ALIGN = 128   # elements: every operand starts 256-byte aligned

    def _pack_layout(self):
        """[(name, block index or None, shape)] in blob order - derived from the hyper-parameters only."""
        D, Dc, depth = self.hidden_size, self.condition_channels, self.depth
        Hm = self.blocks[0].mlp.fc1.out_features if depth else 0
        proj_bias = depth > 0 and self.blocks[0].attn.proj.bias is not None
        items = []
        for i in range(depth):
            items += [("w_q", i, (D, D)), ("b_q", i, (D,)), ("w_cproj", i, (D, D)), ("b_cproj", i, (D,) if proj_bias else None),
                      ("w_qkv", i, (3 * D, D)), ("b_qkv", i, (3 * D,)), ("w_proj", i, (D, D)),
                      ("b_proj", i, (D,) if proj_bias else None), ("w_fc1", i, (Hm, D)), ("b_fc1", i, (Hm,)),
                      ("w_fc2", i, (D, Hm)), ("b_fc2", i, (D,))]
        # to_k / to_v of ALL blocks read the same conditioning tokens: one [depth*2D, Dc] matrix; ONE adaLN matrix
        items += [("w_kv_all", None, (depth * 2 * D, Dc) if depth else None), ("b_kv_all", None, (depth * 2 * D,) if depth else None),
                  ("w_ada", None, (depth * 9 * D + 2 * D, D)), ("b_ada", None, (depth * 9 * D + 2 * D,)),
                  ("w_final", None, (self.out_channels, D)), ("b_final", None, (self.out_channels,))]
        return items

    def packed_alloc(self, dtype: torch.dtype, device=None) -> Dict:
        """Allocate the blob and the operand views WITHOUT filling them (a non-source rank of a broadcast does this)."""
        device = self.x_embedder.weight.device if device is None else torch.device(device)
        layout = self._pack_layout()
        al = self._PACK_ALIGN
        total = sum(ops.round_up(math.prod(shp), al) for _, _, shp in layout if shp is not None)
        flat = torch.zeros(total, dtype=dtype, device=device)      # alignment gaps stay zero: the blob is reproducible byte for byte
        pk: Dict = {"blocks": [dict() for _ in range(self.depth)], "_flat": flat}
        off = 0
        for name, blk, shp in layout:
            view = None
            if shp is not None:
                n = math.prod(shp)
                view = flat[off:off + n].view(shp)
                off += ops.round_up(n, al)
            (pk if blk is None else pk["blocks"][blk])[name] = view
        self._pack = {(dtype, device): pk}
        self._cond = None
        return pk

    def packed(self, dtype: torch.dtype) -> Dict:
        """One-time conversion of the fp32 parameters into the 16-bit operands the kernels stream:
        per block fused [to_k; to_v] and qkv matrices, and ONE adaLN matrix for all blocks + final."""
        dev = self.x_embedder.weight.device
        key = (dtype, dev)
        if key in self._pack:
            return self._pack[key]
        if self._packed_only:
            raise RuntimeError(f"this module holds only packed 16-bit weights (broadcast_packed_ / pack_from_state_dict / "
                               f"load_packed), and not for ({dtype}, {dev}): the fp32 parameters were never loaded")
        pk = self.packed_alloc(dtype, dev)
        self._fill_pack(pk, lambda prm: prm)
        return pk

    def _fill_pack(self, pk: Dict, src) -> None:
        """Fill the blob's operand views; `src(parameter)` returns the tensor holding that parameter's values (the
        parameter itself, or the checkpoint's entry of the same name).  The cast to the blob's dtype happens in the copy."""
        D = self.hidden_size
        with torch.no_grad():
            for i, blk in enumerate(self.blocks):
                ca, sa, mlp, w = blk.crossattn, blk.attn, blk.mlp, pk["blocks"][i]
                for name, lin in (("q", ca.to_q), ("cproj", ca.proj), ("qkv", sa.qkv), ("proj", sa.proj), ("fc1", mlp.fc1),
                                  ("fc2", mlp.fc2)):
                    w["w_" + name].copy_(src(lin.weight))
                    if w["b_" + name] is not None:
                        w["b_" + name].copy_(src(lin.bias))
                pk["w_kv_all"][i * 2 * D:i * 2 * D + D].copy_(src(ca.to_k.weight))
                pk["w_kv_all"][i * 2 * D + D:(i + 1) * 2 * D].copy_(src(ca.to_v.weight))
                pk["b_kv_all"][i * 2 * D:i * 2 * D + D].copy_(src(ca.to_k.bias))
                pk["b_kv_all"][i * 2 * D + D:(i + 1) * 2 * D].copy_(src(ca.to_v.bias))
                pk["w_ada"][i * 9 * D:(i + 1) * 9 * D].copy_(src(blk.adaLN_modulation[1].weight))
                pk["b_ada"][i * 9 * D:(i + 1) * 9 * D].copy_(src(blk.adaLN_modulation[1].bias))
            base = self.depth * 9 * D
            pk["w_ada"][base:].copy_(src(self.final_layer.adaLN_modulation[1].weight))
            pk["b_ada"][base:].copy_(src(self.final_layer.adaLN_modulation[1].bias))
            pk["w_final"].copy_(src(self.final_layer.linear.weight))
            pk["b_final"].copy_(src(self.final_layer.linear.bias))

    # ---- checkpoint -> packed blob without the fp32 round trip.  The reference loads its fp16
    # `.pt` into fp32 parameters (inference.py:257-259: 3.6 GB) and autocast re-casts them on every use; here the
    # checkpoint's tensors are copied straight into the 16-bit blob the kernels read, and the blob can be written to /
    # mapped from ONE flat file so that a later start is a single host-to-device copy of 1.82 GB.
    def pack_from_state_dict(self, state_dict: Dict, dtype: torch.dtype, device=None) -> Dict:
        """Strict like `load_state_dict(strict=True)`: every parameter name must be present with its shape, nothing else.
        The large matrices go only into the blob (this module becomes packed-only: its fp32 route raises); the few
        fp32 tensors used outside autocast (embedders, null conditioning row) are loaded into their parameters."""
        names = {id(prm): n for n, prm in self.named_parameters()}
        missing = [n for n in names.values() if n not in state_dict]
        known = set(names.values()) | {n for n, _ in self.named_buffers()}      # (buffers - PointEmbed.basis - are constants)
        extra = [k for k in state_dict if k not in known]
        if missing or extra:
            raise RuntimeError(f"pack_from_state_dict: missing keys {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                               f"unexpected keys {extra[:4]}{'...' if len(extra) > 4 else ''}")
        for prm in self.parameters():
            if tuple(state_dict[names[id(prm)]].shape) != tuple(prm.shape):
                raise RuntimeError(f"pack_from_state_dict: shape mismatch for {names[id(prm)]}: checkpoint "
                                   f"{tuple(state_dict[names[id(prm)]].shape)}, model {tuple(prm.shape)}")
        self.repack()
        pk = self.packed_alloc(dtype, device)
        self._fill_pack(pk, lambda prm: state_dict[names[id(prm)]])
        with torch.no_grad():
            for t, prm in zip(self.small_fp32_tensors(), self._small_fp32_params()):
                t.copy_(state_dict[names[id(prm)]])
        self._packed_only = True
        return pk

    PACKED_MAGIC = b"PRIMXPK1"
    # header "format": 2 since round 3 (`hyper` gained the class name and DiTAdditivePosEmb's fp32 tensor list grew: files of
    # format 1 have another layout behind the blob and must be re-packed)
    PACKED_FORMAT = 2
    _PACKED_FILE_ALIGN = 4096

    def _hyper(self) -> Dict:
        proj_bias = self.depth > 0 and self.blocks[0].attn.proj.bias is not None
        return {"class": type(self).__name__, "seq_length": self.seq_length, "in_channels": self.in_channels, "condition_channels": self.condition_channels,
                "hidden_size": self.hidden_size, "depth": self.depth, "num_heads": self.num_heads,
                "mlp_hidden": self.blocks[0].mlp.fc1.out_features if self.depth else 0, "attn_proj_bias": bool(proj_bias),
                "cond_drop_prob_positive": self.cond_drop_prob > 0, "out_channels": self.out_channels}

    def save_packed(self, path: str, dtype: torch.dtype) -> int:
        """Write the packed blob + the small fp32 tensors as ONE flat file:
        magic (8 bytes) | header length (u64 LE) | JSON header | zero pad to 4096 | blob | zero pad | fp32 tensors.
        The header records the hyper-parameters the blob layout is derived from; `load_packed` refuses a mismatch.
        Returns the file size."""
        import json
        pk = self.packed(dtype)
        flat = pk["_flat"].detach().cpu().contiguous()
        small = [t.detach().cpu().float().contiguous() for t in self.small_fp32_tensors()]
        al = self._PACKED_FILE_ALIGN
        head = {"format": self.PACKED_FORMAT, "dtype": str(dtype).replace("torch.", ""), "hyper": self._hyper(), "pack_align": self._PACK_ALIGN,
                "blob_elements": flat.numel(),
                "small_fp32": [list(t.shape) for t in small]}
        hj = json.dumps(head, sort_keys=True).encode()
        blob_off = ops.round_up(16 + len(hj), al)
        blob_bytes = flat.numel() * flat.element_size()
        small_off = ops.round_up(blob_off + blob_bytes, al)
        with open(path, "wb") as f:
            f.write(self.PACKED_MAGIC)
            f.write(len(hj).to_bytes(8, "little"))
            f.write(hj)
            f.write(b"\0" * (blob_off - 16 - len(hj)))
            raw = flat.view(torch.int16).numpy()
            for lo in range(0, raw.size, 1 << 25):      # 64 MiB pieces: no second 1.82 GB bytes object
                f.write(raw[lo:lo + (1 << 25)].tobytes())
            f.write(b"\0" * (small_off - blob_off - blob_bytes))
            for t in small:
                f.write(t.numpy().tobytes())
            return f.tell()

    def load_packed(self, path: str, device=None) -> Dict:
        """Map a `save_packed` file and copy it into a freshly allocated blob on `device` (one host-to-device copy straight
        from the page cache: no fp32 parameters, no per-layer casts).  This module becomes packed-only."""
        import json
        import os
        with open(path, "rb") as f:
            if f.read(8) != self.PACKED_MAGIC:
                raise RuntimeError(f"{path}: not a packed PrimX DiT file")
            hlen = int.from_bytes(f.read(8), "little")
            head = json.loads(f.read(hlen).decode())
        if head.get("format") != self.PACKED_FORMAT:
            raise RuntimeError(f"{path}: packed file format {head.get('format')} (this build reads format {self.PACKED_FORMAT}): "
                               "repack required - write it again with DiT.save_packed from the checkpoint")
        if head.get("hyper") != self._hyper() or head.get("pack_align") != self._PACK_ALIGN:
            raise RuntimeError(f"{path}: packed for a different model ({head.get('hyper')}) than this one ({self._hyper()})")
        dtype = {"float16": torch.float16, "bfloat16": torch.bfloat16}[head["dtype"]]
        self.repack()
        pk = self.packed_alloc(dtype, device)
        flat = pk["_flat"]
        if flat.numel() != head["blob_elements"]:
            raise RuntimeError(f"{path}: blob has {head['blob_elements']} elements, this model's layout {flat.numel()}")
        al = self._PACKED_FILE_ALIGN
        blob_off = ops.round_up(16 + hlen, al)
        blob_bytes = flat.numel() * flat.element_size()
        small_off = ops.round_up(blob_off + blob_bytes, al)
        mm = torch.from_file(path, shared=False, size=os.path.getsize(path), dtype=torch.uint8)   # private mapping of the file
        with torch.no_grad():
            flat.view(torch.int16).copy_(mm[blob_off:blob_off + blob_bytes].view(torch.int16))
            off = small_off
            for t, shp in zip(self.small_fp32_tensors(), head["small_fp32"]):
                if list(t.shape) != shp:
                    raise RuntimeError(f"{path}: fp32 tensor of shape {shp} does not fit {list(t.shape)}")
                n = t.numel()
                t.copy_(mm[off:off + 4 * n].view(torch.float32).view(shp))
                off += 4 * n
        del mm
        self._packed_only = True
        return pk

    def _small_fp32_params(self):
        ps = [self.x_embedder.weight, self.x_embedder.bias, self.t_embedder.mlp[0].weight, self.t_embedder.mlp[0].bias,
              self.t_embedder.mlp[2].weight, self.t_embedder.mlp[2].bias]
        if self.cond_drop_prob > 0:
            ps.append(self.null_cond_embedding)
        return ps

    def small_fp32_tensors(self):
        """The fp32 parameters the 16-bit path reads directly (embedders outside autocast, dit_crossattn.py:191-192, and
        the null conditioning row): what travels next to the packed blob in a broadcast."""
        return [t.data for t in self._small_fp32_params()]

    def _heads(self, tag: str, B: int, n: int, kind: int, dtype, device, pad_to: int) -> torch.Tensor:
        """Persistent zero-padded attention operand buffers (pads are never written, so they stay zero)."""
        key = (tag, B, n, kind, dtype, str(device), pad_to)
        buf = self._heads_ws.get(key)
        if buf is None:
            buf = ops.alloc_heads(B, self.num_heads, n, self.hidden_size // self.num_heads, kind, dtype, device, pad_to,
                                  role=tag[0].lower())  # "q" / "k": operand-level key-padding mask (ops.alloc_heads)
            self._heads_ws[key] = buf
        # every shape group that uses a workspace holds it (keys that do not depend on the batch size - the broadcast entries
        # Kn / Vn - are shared between groups): it is freed when the LAST of them is evicted, not when the first allocator is
        self._heads_owner.setdefault(key, set()).add(self._heads_group)
        return buf

    _HEADS_GROUPS = 3   # (batch, N, L, dtype, device) shapes whose attention workspaces stay allocated

    def _heads_begin(self, group) -> None:
        """Called once at the start of a forward: marks `group` most recently used and, when more than `_HEADS_GROUPS` shapes
        are alive, frees the workspaces of the least recently used ones (never in the middle of a forward; a model that
        alternates between a few shapes keeps all of them instead of re-allocating hundreds of MB per call)."""
        lru = self._heads_lru
        if group in lru:
            lru.remove(group)
        lru.append(group)
        self._heads_group = group
        while len(lru) > self._HEADS_GROUPS:
            old = lru.pop(0)
            for k in [k for k, gs in self._heads_owner.items() if old in gs]:
                gs = self._heads_owner[k]
                gs.discard(old)
                if not gs:
                    self._heads_ws.pop(k, None)
                    self._heads_owner.pop(k, None)
            # (the conditioning cache `_cond` always belongs to a more recently used group than the evicted one - it is rebuilt or
            # re-tagged by every forward - and validates its K / V buffers by pointer (`kv_id`), so there is nothing to drop here)

    def _side_stream(self, dev) -> "torch.cuda.Stream":
        key = str(dev)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=dev)
        return self._side[key]

    # ------------------------------------------------------------------ conditioning (step-invariant inputs)
    def _cond_state(self, y: torch.Tensor, null_half: bool, dt) -> Dict:
        """16-bit image of the conditioning tokens as the K / V projection reads them, cached per conditioning tensor.

        What the reference does every step - `cat([y, y_null])` (dit_crossattn.py:207-208) and the autocast cast in
        front of to_k / to_v (attention.py:106-107) - depends only on `y`, so it is done once per `y` and reused while
        the caller keeps passing the same tensor (same storage, shape, strides and in-place version; the cache holds a
        reference to `y`, so its storage cannot be recycled for another tensor behind our back).  This is a cast of an
        INPUT, not skipped arithmetic.  The step-invariant K / V projections themselves are reused only when
        `self.reuse_cond_kv` is set (exact algebra; off by default so that every step projects
        its conditioning tokens, as the reference does)."""
        B, L, Dc = y.shape
        Be = 2 * B if null_half else B
        st = self._cond
        # (a tensor created under torch.inference_mode() tracks no version: in-place edits of it could not be detected, so such a
        # `y` is never served from the cache - its 16-bit image is rebuilt on every call)
        ver = None if y.is_inference() else y._version
        if st is not None and ver is not None:
            k = st["y"]
            if (k.data_ptr() == y.data_ptr() and k.shape == y.shape and k.stride() == y.stride() and k.dtype == y.dtype
                    and st["ver"] == ver and st["dt"] == dt and st["null_half"] == null_half):
                return st
        # Conditioning rows per batch entry as the K / V projection sees them: padded with zero rows to a multiple of 256
        # when that costs <= 12.5 % (1370 -> 1536), so that GEMM tiles never straddle batch entries and the projection
        # takes the 256x288 tile (csrc/gemm.hip launch()).  The pad rows produce K = bias_k / V = bias_v entries beyond
        # the L valid keys: the attention kernel never visits tiles past L, and inside the last tile they carry the
        # operand-level key mask (ops.alloc_heads), i.e. probability exactly 0.
        Lk = L
        if (ops.round_up(L, 256) - L) * 8 <= L:
            Lk = ops.round_up(L, 256)
        buf = torch.zeros(Be, Lk, Dc, dtype=dt, device=y.device)
        yf = y.float().contiguous()
        for b in range(B):
            ops.cast16(yf[b], dt, out=buf[b, :L])
        if null_half:   # y_null = null_cond_embedding.expand_as(y)  (dit_crossattn.py:207): one row, broadcast once
            null16 = ops.cast16(self.null_cond_embedding.detach().float().contiguous().to(y.device), dt)
            buf[B:, :L] = null16
        st = {"y": y, "ver": ver, "dt": dt, "null_half": null_half, "Lk": Lk, "y16": buf.view(Be * Lk, Dc),
              "kv_valid": False, "null16": None}
        if null_half and L >= 64:
            # the broadcast operand entry of the unconditional half: ops.bcast_keys(L) copies of the null row (one full 64-key tile
            # + the ragged last tile of the L-key sequence)
            st["null16"] = null16.reshape(1, Dc).expand(ops.bcast_keys(L), Dc).contiguous()
        self._cond = st
        return st

    # ------------------------------------------------------------------ forward
    @ops.on_input_device
    def forward(self, x, t, y, precision_dtype=torch.float32, enable_amp=False):
        """x: (B, N, C) fp32; t: (B,) int; y: (B, L, Dc) fp32 -> (B, N, out_channels).

        `enable_amp=True` with fp16 / bf16: the autocast topology on the 16-bit MFMA path, output in `precision_dtype`.
        `enable_amp=False` (the signature default; `precision: tf32` of the CLI, inference.py:239-247) or
        `precision_dtype=float32`: exact fp32 on the fp32 matrix instruction (csrc/fp32.hip), fp32 output."""
        self._check_inputs(x)
        if not enable_amp or precision_dtype == torch.float32:
            return self._forward_fp32(x, t, y)
        if precision_dtype not in (torch.float16, torch.bfloat16):
            raise NotImplementedError(f"autocast dtype {precision_dtype} is not supported (fp16 / bf16 / fp32)")
        return self._forward16(x, t, y, precision_dtype, null_half=False)

    def _check_inputs(self, x):
        if self.training:
            raise NotImplementedError("the accelerated DiT is inference-only: call .eval()")
        if not x.is_cuda:
            raise RuntimeError("DiT.forward needs HIP device tensors; there is no CPU path")

    # ---- timestep plan.  The adaLN modulation of all blocks (dit_crossattn.py:40-43,54,69-75) depends on the timestep only,
    # and a sampling loop knows its timesteps in advance: computed per step it is a 674 MB weight stream for ONE distinct input
    # row (140 us of every 9.6 ms step at configs[1]); computed per LOOP the same rows go through the same kernel eight at a time
    # (ceil(n / 8) streams per loop instead of n).  Every row is still computed, by the same arithmetic: the few-row kernels
    # (`gemv_f32_kernel`, `gemv16_kernel`) give a row the same bits alone or in a batch, so planned and unplanned loops produce
    # identical samples (tests/test_hip_dit.py).
    _PLAN_ROWS = 8   # rows per pass of the few-row kernels (csrc/gemm.hip GEMV_MAX_ROWS16, csrc/rowops.hip GEMV_MAX_ROWS)

    def plan_timesteps(self, timesteps: torch.Tensor) -> None:
        """`timesteps`: 1-D int64 device tensor - the model timestep of each coming call, every batch entry of a call sharing
        it.  The caller then sets `select_planned_timestep(i)` before call i (the sampler does both); a forward without a
        selected row computes its modulation from `t` as usual."""
        if timesteps.dim() != 1 or timesteps.dtype != torch.int64 or not timesteps.is_cuda:
            raise RuntimeError("plan_timesteps: expected a 1-D int64 HIP device tensor")
        self._t_plan = {"t": timesteps.contiguous(), "mod": {}, "row": None}

    def select_planned_timestep(self, row: Optional[int]) -> None:
        if self._t_plan is not None:
            if row is not None and not (0 <= row < self._t_plan["t"].numel()):
                raise IndexError(f"planned timestep {row} out of range")
            self._t_plan["row"] = row

    def clear_timestep_plan(self) -> None:
        self._t_plan = None                   # (drops the loop's tables: modulation rows and the fold's u / v rows)
        self._fold_fp16_used = False

    def _modulation_table(self, plan: Dict, dt: torch.dtype, pk: Dict) -> torch.Tensor:
        key = (dt, pk["w_ada"].data_ptr())
        tab = plan["mod"].get(key)
        if tab is None:
            ts = plan["t"]
            n, R = ts.numel(), self._PLAN_ROWS
            tab = torch.empty(n, pk["w_ada"].shape[0], dtype=dt, device=ts.device)
            for lo in range(0, n, R):
                st16 = ops.silu_cast(self.t_embedder(ts[lo:lo + R]), dt)
                ops.linear(st16, pk["w_ada"], pk["b_ada"], out=tab[lo:lo + R])
            plan["mod"] = {key: tab}
        return tab

    def _fold_ok(self, T: int, N: int) -> bool:
        return bool(self.fold_ln) and self.depth > 0 and ops.fold_shapes_ok(T, N, self.hidden_size, self.num_heads)

    def _fold_tables(self, plan: Dict, dt: torch.dtype, pk: Dict) -> Dict:
        """The fold's per-timestep vectors of a planned loop: for every block and site (to_q, qkv, fc1) u = cast16(1 + scale) W^T
        and v = shift W^T + b in fp32, all planned timesteps in ONE GEMM per (block, site) - 3 x depth launches per loop, the rows
        [2, n_timesteps, N_site]."""
        key = (dt, pk["w_ada"].data_ptr())
        ft = plan.get("fold")
        if ft is not None and ft["key"] == key:
            return ft
        tab = self._modulation_table(plan, dt, pk)
        n, D = tab.shape[0], self.hidden_size
        A = self._fold_rows(tab, self.depth, D)
        # (3 x depth - 1 independent few-row GEMMs.  As launches of their own: 9 - 36 workgroups and ~20 us each, 0.07 ms per step of a
        # 25-step loop; issued round-robin on four side streams they overlapped - and every LATER launch of the loop got slower: the step
        # 8.83 vs 8.63 ms on the same box (profiles/r4_experiments.txt section 6).  Round 6: one grouped launch on the calling stream.)
        # ONE grouped launch for the 3 x depth - 1 sites (primx_linear_f32out_group, ABI 26: 592 MB of weights streamed once; the per-site
        # launches were 8 - 32 workgroups and ~18 us each = 1.5 ms per loop); per-site launches where the grouped kernel does not apply
        sites, uv = [], []
        total = sum(2 * n * w["w_" + name].shape[0] for i, w in enumerate(pk["blocks"]) for s, name in enumerate(("q", "qkv", "fc1"))
                    if not (i == 0 and s == 0))
        flat, off = torch.empty(total, dtype=torch.float32, device=tab.device), 0
        for i, w in enumerate(pk["blocks"]):
            row = []
            for s, name in enumerate(("q", "qkv", "fc1")):
                if i == 0 and s == 0:                                     # (the first LayerNorm of a forward stays a launch)
                    row.append(None)
                    continue
                Ns = w["w_" + name].shape[0]
                out = flat[off:off + 2 * n * Ns].view(2, n, Ns)
                off += 2 * n * Ns
                sites.append((A[i, s].view(2 * n, D), w["w_" + name], w["b_" + name], out.view(2 * n, Ns)))
                row.append(out)
            uv.append(row)
        if not ops.linear_f32out_group(sites, bias_from_row=n):
            for a_s, w_s, b_s, o_s in sites:
                ops.linear_f32out(a_s, w_s, b_s, o_s, bias_from_row=n)
        ft = plan["fold"] = {"key": key, "uv": uv}
        return ft

    def fold_overflowed(self, sample: torch.Tensor) -> bool:
        """Asked by the sampling loop about its FINAL sample (diffusion/sampler.py `_fold_guard`): did forwards of this loop run the
        LayerNorm fold in fp16 AND is the sample non-finite?  (One reduction + one read-back per loop, only when the fold ran in
        fp16.)  The sampler then repeats the loop with LayerNorm launches - see `fold_ln` in __init__ for why this is not expected
        to happen."""
        used, self._fold_fp16_used = self._fold_fp16_used, False
        return bool(used) and not bool(torch.isfinite(sample).all())

    @staticmethod
    def _fold_rows(tab: torch.Tensor, depth: int, D: int) -> torch.Tensor:
        """The A operands of the fold's u / v GEMMs from the loop's modulation table `tab` [n_timesteps, depth * 9 D (+ 2 D)] (16-bit;
        per block: shift, scale, gate of the cross-attention, self-attention and MLP branch - dit_crossattn.py:54):
        A[block, site, 0] = cast16(1 + scale) and A[block, site, 1] = shift of every timestep, [depth, 3, 2, n_timesteps, D]."""
        n = tab.shape[0]
        v = tab[:, :depth * 9 * D].view(n, depth, 3, 3, D)               # [timestep, block, site, (shift, scale, gate), D]
        A = torch.empty(depth, 3, 2, n, D, dtype=tab.dtype, device=tab.device)
        A[:, :, 0] = (1 + v[:, :, :, 1]).permute(1, 2, 0, 3)              # (1 + scale) is formed in the 16-bit type, as autocast does
        A[:, :, 1] = v[:, :, :, 0].permute(1, 2, 0, 3)
        return A

    def _forward16(self, x, t, y, dt, null_half: bool):
        """The 16-bit autocast path.  `null_half`: classifier-free guidance - the effective batch is [x; x] with the
        second half conditioned on the null embedding (dit_crossattn.py:204-209), assembled here without materialising
        the concatenations: tokens are embedded into both halves of the residual stream, t is embedded once."""
        B, N, Cin = x.shape
        Be = 2 * B if null_half else B
        L, Dc = y.shape[1], y.shape[2]
        D, H = self.hidden_size, self.num_heads
        dh = D // H
        T = Be * N
        pk = self.packed(dt)
        dev = x.device
        self._heads_begin((Be, N, L, dt, str(dev)))

        xf = x.reshape(B * N, Cin).float().contiguous()
        h = torch.empty(T, D, dtype=torch.float32, device=dev)
        # cat([x, x]) embeds to the same rows twice (dit_crossattn.py:205,191): the embedding kernel writes both halves
        self._embed_tokens(xf, h[:B * N], h[B * N:] if null_half else None)
        plan = self._t_plan
        if plan is not None and plan["row"] is not None:
            # the sampling loop announced its timesteps (plan_timesteps): this call's modulation is a row of the per-loop
            # table, shared by all batch entries (row stride 0)
            mod = self._modulation_table(plan, dt, pk)[plan["row"]:plan["row"] + 1].expand(Be, -1)
        else:
            t_emb = self.t_embedder(t)                               # [B, D]
            st16 = torch.empty(Be, D, dtype=dt, device=dev)
            ops.silu_cast(t_emb, dt, out=st16[:B])
            if null_half:
                ops.silu_cast(t_emb, dt, out=st16[B:])
            # adaLN for every block + final layer: SiLU -> one streaming GEMM (dit_crossattn.py:40-43,54,69-75)
            mod = ops.linear(st16, pk["w_ada"], pk["b_ada"])         # [Be, depth*9D + 2D]
        cs = self._cond_state(y, null_half, dt)
        cs["group"] = self._heads_group
        Lk, y16 = cs["Lk"], cs["y16"]

        nq_pad = ops.round_up(N, ops.BQ)
        Qc = self._heads("Qc", Be, N, HEADS_ROWS, dt, dev, ops.BQ)
        # cross-attention K / V of every block in ONE projection GEMM (N = depth * 2D): [depth*Be, H, L_pad, DP]
        kv_pad = 256 if Lk != L else ops.BKV
        # (dedup: the operands hold the B conditional entries only; the unconditional ones share the broadcast entry Kn / Vn)
        dedup = bool(self.dedup_null_kv) and null_half and cs["null16"] is not None and self.depth > 0
        Bkv = B if dedup else Be
        Kc = self._heads("Kc", self.depth * Bkv, L, HEADS_KROWS, dt, dev, kv_pad)
        Vc = self._heads("Vc", self.depth * Bkv, L, HEADS_VT, dt, dev, kv_pad)
        Kc_blk = Kc.view(self.depth, Bkv, *Kc.shape[1:])
        Vc_blk = Vc.view(self.depth, Bkv, *Vc.shape[1:])
        Kn_blk = Vn_blk = None
        if dedup:
            Ln = ops.bcast_keys(L)
            Kn = self._heads("Kn", self.depth, Ln, HEADS_KROWS, dt, dev, ops.BKV)
            Vn = self._heads("Vn", self.depth, Ln, HEADS_VT, dt, dev, ops.BKV)
            Kn_blk, Vn_blk = Kn.view(self.depth, 1, *Kn.shape[1:]), Vn.view(self.depth, 1, *Vn.shape[1:])
        need_kv = bool(self.depth) and not (self.reuse_cond_kv and cs["kv_valid"] and cs.get("kv_id") == (Kc.data_ptr(), Vc.data_ptr()))

        ride_state = {"on": False}

        def kv_problem(i: int):
            # block i's [to_k; to_v] projection of the conditional entries as the heads GEMM primx_linear_heads_fold_pair carries
            wkv, bkv = pk["w_kv_all"][i * 2 * D:(i + 1) * 2 * D], pk["b_kv_all"][i * 2 * D:(i + 1) * 2 * D]
            return (y16[:Bkv * Lk], wkv, bkv, Lk, H, dh, [HEADS_KROWS, HEADS_VT], [Kc_blk[i], Vc_blk[i]], Kc.shape[2], 1.0, Bkv * L)

        def project_kv(cond_rows: bool) -> None:
            # (cond_rows False: the one-call route projects the conditional entries itself - `kv_ride`)
            if cond_rows:
                ops.linear_heads(y16[:Bkv * Lk], pk["w_kv_all"], pk["b_kv_all"], Lk, H, dh, [HEADS_KROWS, HEADS_VT], [Kc, Vc],
                                 Kc.shape[2], n_rep=self.depth, rep_batches=Bkv, real_rows=Bkv * L)
            if dedup:
                ops.linear_heads(cs["null16"], pk["w_kv_all"], pk["b_kv_all"], Ln, H, dh, [HEADS_KROWS, HEADS_VT], [Kn, Vn],
                                 Kn.shape[2], n_rep=self.depth, rep_batches=1)
            cs["kv_valid"], cs["kv_id"] = True, (Kc.data_ptr(), Vc.data_ptr())
        Qs = self._heads("Qs", Be, N, HEADS_ROWS, dt, dev, ops.BQ)
        Ks = self._heads("Ks", Be, N, HEADS_KROWS, dt, dev, ops.BQ)
        Vs = self._heads("Vs", Be, N, HEADS_VT, dt, dev, ops.BQ)
        xn = torch.empty(T, D, dtype=dt, device=dev)
        att = torch.empty(Be, N, D, dtype=dt, device=dev)
        hid = torch.empty(T, pk["blocks"][0]["w_fc1"].shape[0], dtype=dt, device=dev) if self.depth else None
        scale = dh ** -0.5

        # Weight prefetch (PRIMX_WPREFETCH=0 turns it off): the 1.8 GB of weights stream through HBM once per forward, and the
        # 128 x 144 GEMMs run 2 - 4 us longer with cold weights than with cache-resident ones (rocprofv3, tools/gpu/r3_touch.sh).
        # `weight_prefetch` = 1: every LayerNorm launch - a short kernel that reads the residual stream from the Infinity Cache -
        # carries the prefetch of the weights of the loader-wave GEMMs that follow it (ops.layernorm_modulate `prefetch=`): no
        # launch, no event, no stream of its own.  `weight_prefetch` = 2 (the default): the GEMM launches carry it instead (`carry=`):
        # the compute waves of a loader-wave kernel touch the lines of a LATER GEMM's weights in front of their k-loop - to_q
        # carries cproj's weights, cproj proj's, fc1 fc2's, fc2 the next block's to_q - and the LayerNorms keep to their own bytes.
        wpf = int(self.weight_prefetch)
        collapse = bool(self.collapse_null_cross_attention) and null_half
        blocks = pk["blocks"]
        # `fuse_ln` (round 4): every gated residual add of a block is followed by the LayerNorm + modulate of the next branch
        # (dit_crossattn.py:55-57) - of the next block after fc2, of the final layer after the last one - so both are requested
        # from one entry point, ops.linear_gate_residual(ln=...): bit-identical to the separate calls, and only the FIRST LayerNorm
        # of a forward is a call of its own.  Whether the library runs the LayerNorm as a second launch (default) or in the tail of
        # the GEMM kernel is decided by the `sync` words (`ln_in_kernel`).  Not with the LayerNorm-carried prefetch (wpf == 1: that
        # mode needs the LayerNorm launches to carry the ranges).
        fuse = bool(self.fuse_ln) and wpf != 1
        sync = None
        sync_w = ops.ln_sync_words(T)
        if fuse and self.ln_in_kernel:
            # (two regions: with `cfg_streams` the two halves run concurrently and must not share words)
            key = str(dev)
            sync = self._ln_sync.get(key)
            if sync is None or sync.numel() < 2 * sync_w:
                sync = self._ln_sync[key] = torch.zeros(2 * sync_w, dtype=torch.int32, device=dev)
        base = self.depth * 9 * D
        fin_mod = (mod[:, base:base + D], mod[:, base + D:base + 2 * D])    # final layer's shift / scale
        # the LayerNorm fold (`fold_ln`): planned calls only - u / v come from the loop's tables, shared by all batch entries
        fold_uv = fcent = fpart = None
        prow = plan["row"] if plan is not None else None
        if (prow is not None and fuse and not collapse and not (self.cfg_streams and null_half and ops.PROFILE is None)
                and plan["t"].numel() <= self.fold_max_steps and self._fold_ok(T, N)):
            fold_uv = self._fold_tables(plan, dt, pk)["uv"]
            if dt == torch.float16:
                self._fold_fp16_used = True
            wk = (str(dev), T)
            if wk not in self._fold_ws:
                self._fold_ws = {wk: ops.fold_workspace(T, D, dev)}
            fcent, fpart = self._fold_ws[wk]

        fside: Dict[int, int] = {}                       # per row range (its b0): which of the fold's two (centre, scale) arrays the current site reads

        def warm(*wts):
            return wts if wpf == 1 else ()

        def carry(wt):
            return wt if wpf == 2 else None

        def block(i, w, b0, b1, hook=None):
            """DiTBlock i on batch entries [b0, b1) (dit_crossattn.py:51-58): 11 launches on the current stream, 8 with the LayerNorm fold."""
            r0, r1 = b0 * N, b1 * N
            hh, xh, ah = h[r0:r1], xn[r0:r1], att[b0:b1]
            Th = r1 - r0
            m = mod[b0:b1, i * 9 * D:(i + 1) * 9 * D]
            ch = [m[:, j * D:(j + 1) * D] for j in range(9)]  # shift/scale/gate x (mca, msa, mlp)
            def ln_of(shift, scale):                     # the LayerNorm that follows a gated residual add, fused into its GEMM
                if not fuse:
                    return None
                return (shift, scale, xh, self.LN_EPS, None if sync is None else (sync[:sync_w] if b0 == 0 else sync[sync_w:]))
            # ---- cross-attention to the image tokens (dit_crossattn.py:55, attention.py:96-114)
            folded = fold_uv is not None
            fp = fpart[r0:r1] if folded else None
            uvs = fold_uv[i] if folded else None

            def fc_in():                                 # the (centre, scale) pairs the current site's producer used ...
                return fcent[fside[b0]][r0:r1]

            def fc_flip():                               # ... and where its consumer leaves the next site's: the other array
                fside[b0] ^= 1
                return fcent[fside[b0]][r0:r1]

            def uv(s):                                   # this call's (u, v) of site s: rows of the planned loop's tables
                return uvs[s][0, prow], uvs[s][1, prow]
            if not fuse or i == 0:                       # (fused: the previous block's fc2 launch has normalised these rows)
                # (the block's cross-attention K / V as the prefetch instead: -1.0 us on that kernel, +0.5 on this one)
                ops.layernorm_modulate(hh, ch[0], ch[1], N, xh, self.LN_EPS, prefetch=warm(w["w_q"], w["w_cproj"]))
                if folded:
                    fside[b0] = 0
                    ops.row_stats(hh, self.LN_EPS, fc_in())   # (centre, scale) of the first folded site: this LayerNorm's (mean, rstd)
            bc = min(b1, B) if collapse else b1          # batch entries [b0, bc) attend; [bc, b1) are unconditional rows
            if bc > b0:
                if folded and i > 0:                     # (folded: bc == b1, every row attends)
                    ops.linear_heads_fold(xh, w["w_q"], N, H, dh, [HEADS_ROWS], [Qc[b0:bc]], nq_pad, fp, *uv(0), fc_in(), fc_flip(),
                                          self.LN_EPS, scale0=scale, carry=carry(w["w_cproj"]))
                else:
                    ops.linear_heads(xh[:(bc - b0) * N], w["w_q"], w["b_q"], N, H, dh, [HEADS_ROWS], [Qc[b0:bc]], nq_pad,
                                     scale0=scale, carry=carry(w["w_cproj"]))
                if dedup and bc > B:                     # entries [max(b0, B), bc) take the broadcast key / value entry
                    kc = Kc_blk[i][b0:B] if b0 < B else None
                    ops.attention(Qc[b0:bc], kc, Vc_blk[i][b0:B] if b0 < B else None, N, L, dh, scale, out=ah[:bc - b0],
                                  bcast=(Kn_blk[i], Vn_blk[i]))
                else:
                    ops.attention(Qc[b0:bc], Kc_blk[i][b0:bc], Vc_blk[i][b0:bc], N, L, dh, scale, out=ah[:bc - b0])
            if bc < b1:
                # V^T layout [b, h, DP, n_pad] (key 0 sits at position 0 of its quad-permuted group): the value row of every head
                nb = b1 - max(bc, b0)
                vsrc = Vn_blk[i].expand(nb, -1, -1, -1) if dedup else Vc_blk[i][max(bc, b0):b1]
                vrow = vsrc[:, :, :dh, 0].reshape(nb, 1, D)
                ah[max(bc, b0) - b0:].copy_(vrow.expand(-1, N, -1))
            if hook is not None:
                hook()
            if folded:                                   # producer of the qkv site
                ops.linear_gate_residual_fold(ah.view(Th, D), w["w_cproj"], w["b_cproj"], ch[2], hh, N, ch[4], fc_in(), xh, fp,
                                              carry=carry(w["w_proj"]))
            else:
                ops.linear_gate_residual(ah.view(Th, D), w["w_cproj"], w["b_cproj"], ch[2], hh, N, carry=carry(w["w_proj"]),
                                         ln=ln_of(ch[3], ch[4]))
            # ---- self-attention over the primitive tokens (dit_crossattn.py:56, attention.py:48-59)
            if not fuse:
                ops.layernorm_modulate(hh, ch[3], ch[4], N, xh, self.LN_EPS, prefetch=warm(w["w_proj"]))
            if folded and ride_state["on"] and i + 1 < len(blocks):
                # (`kv_ride`: the next block's to_k / to_v projection rides on this launch - what primx_dit_blocks_fold issues)
                u1, v1 = uv(1)
                ops.linear_heads_fold_pair(dict(A=xh, W=w["w_qkv"], rows_per_batch=N, heads=H, dh=dh, kinds=[HEADS_ROWS, HEADS_KROWS, HEADS_VT],
                                                dsts=[Qs[b0:b1], Ks[b0:b1], Vs[b0:b1]], n_pad=nq_pad, part=fp, u=u1, v=v1, center=fc_in(),
                                                center_out=fc_flip(), eps=self.LN_EPS), *kv_problem(i + 1))
            elif folded:
                ops.linear_heads_fold(xh, w["w_qkv"], N, H, dh, [HEADS_ROWS, HEADS_KROWS, HEADS_VT],
                                      [Qs[b0:b1], Ks[b0:b1], Vs[b0:b1]], nq_pad, fp, *uv(1), fc_in(), fc_flip(), self.LN_EPS)
            else:
                ops.linear_heads(xh, w["w_qkv"], w["b_qkv"], N, H, dh, [HEADS_ROWS, HEADS_KROWS, HEADS_VT],
                                 [Qs[b0:b1], Ks[b0:b1], Vs[b0:b1]], nq_pad)
            ops.attention(Qs[b0:b1], Ks[b0:b1], Vs[b0:b1], N, N, dh, scale, out=ah)
            if folded:                                   # producer of the fc1 site
                ops.linear_gate_residual_fold(ah.view(Th, D), w["w_proj"], w["b_proj"], ch[5], hh, N, ch[7], fc_in(), xh, fp)
            else:
                ops.linear_gate_residual(ah.view(Th, D), w["w_proj"], w["b_proj"], ch[5], hh, N, ln=ln_of(ch[6], ch[7]))
            # ---- MLP (dit_crossattn.py:57, models/utils.py:94-101)
            if not fuse:
                ops.layernorm_modulate(hh, ch[6], ch[7], N, xh, self.LN_EPS, prefetch=warm(w["w_fc2"]))
            if folded:
                ops.linear_fold(xh, w["w_fc1"], hid[r0:r1], fp, *uv(2), fc_in(), fc_flip(), self.LN_EPS, act=ACT_GELU_TANH,
                                carry=carry(w["w_fc2"]))
            else:
                ops.linear(xh, w["w_fc1"], w["b_fc1"], out=hid[r0:r1], act=ACT_GELU_TANH, carry=carry(w["w_fc2"]))
            last = i + 1 == len(blocks)
            nxt = (fin_mod[0][b0:b1], fin_mod[1][b0:b1]) if last else \
                (mod[b0:b1, (i + 1) * 9 * D:(i + 1) * 9 * D + D], mod[b0:b1, (i + 1) * 9 * D + D:(i + 1) * 9 * D + 2 * D])
            if folded and not last:                      # producer of the next block's to_q site
                ops.linear_gate_residual_fold(hid[r0:r1], w["w_fc2"], w["b_fc2"], ch[8], hh, N, nxt[1], fc_in(), xh, fp,
                                              carry=carry(blocks[i + 1]["w_q"]))
            else:                                        # (the final layer's LayerNorm stays a launch: its Linear has 8 columns)
                ops.linear_gate_residual(hid[r0:r1], w["w_fc2"], w["b_fc2"], ch[8], hh, N,
                                         carry=None if last else carry(blocks[i + 1]["w_q"]), ln=ln_of(*nxt))

        two_streams = bool(self.cfg_streams and null_half and self.depth and ops.PROFILE is None)
        one_call = (not two_streams and fold_uv is not None and self.blocks_call and sync is None and ops.PROFILE is None
                    and self.block_probe is None and _lib.blocks_call_available())
        # `kv_ride`: every folded single-stream forward - the library issues the riders on the one-call route, the block loop below otherwise
        # - where the riders fit the qkv launch's round (T = 4096: 192 + 48 tiles; a large batch fills the chip by itself and keeps the
        # batched projection)
        ride = (need_kv and self.kv_ride and fold_uv is not None and not two_streams and _lib.kv_ride_available()
                and ops.fold_pair_fits(T, N, D, H, Bkv * Lk, Lk, Dc))
        ride_state["on"] = ride and not one_call
        if need_kv:
            project_kv(not ride)
            if ride_state["on"]:
                ops.linear_heads_fold_pair(None, *kv_problem(0))     # block 0's projection: a launch of its own, on the riders' tile kernel
        if self.cfg_streams and null_half and self.depth and ops.PROFILE is None:
            # Two HIP streams, one per CFG half (the conditional and the unconditional rows are independent chains of
            # kernels): every GEMM of this path ends with a write burst that nothing of ITS OWN kernel can overlap
            # (DESIGN_LOG.md section 4); with the second chain a few kernels behind the first, one chain's burst drains
            # under the other chain's matrix phase.  Same kernels, same arithmetic per row.
            main = torch.cuda.current_stream()
            side = self._side_stream(dev)
            side.wait_stream(main)                       # embeddings, modulation and the K / V projection come first
            lag = torch.cuda.Event()
            for i, w in enumerate(pk["blocks"]):
                block(i, w, 0, B, hook=(lambda: lag.record(main)) if i == 0 else None)
                with torch.cuda.stream(side):
                    if i == 0:
                        side.wait_event(lag)             # the side chain starts when the main one is three kernels in
                    block(i, w, B, Be)
            main.wait_stream(side)
        elif one_call:
            # ONE foreign call for the forward's blocks (primx_dit_blocks_fold, ABI 24): the C side issues the launches block() below
            # would - same entry points, same arguments, same order, bit-identical results - at ~1 us of host time each instead of
            # ~21 us of Python + ctypes (tools/host_bound_check.py).  The per-block descriptors are built once per planned loop.
            fd = plan["fold"]
            dkey = (Kc.data_ptr(), Vc.data_ptr(), Kn_blk[0].data_ptr() if dedup else 0, wpf, Bkv)
            if fd.get("desc_key") != dkey:
                arr = (_lib.DitBlockFold * self.depth)()
                rng = lambda t: ops._range(t) if wpf == 2 else (None, 0)
                for i, w in enumerate(blocks):
                    d = arr[i]
                    for name in ("w_q", "b_q", "w_cproj", "b_cproj", "w_qkv", "w_proj", "b_proj", "w_fc1", "w_fc2", "b_fc2"):
                        setattr(d, name, None if w[name] is None else w[name].data_ptr())
                    d.Kc, d.Vc = Kc_blk[i].data_ptr(), Vc_blk[i].data_ptr()
                    d.Kb, d.Vb = (Kn_blk[i].data_ptr(), Vn_blk[i].data_ptr()) if dedup else (None, None)
                    d.uv_q = None if fold_uv[i][0] is None else fold_uv[i][0].data_ptr()
                    d.uv_qkv, d.uv_fc1 = fold_uv[i][1].data_ptr(), fold_uv[i][2].data_ptr()
                    for cname, t in (("carry_q", w["w_cproj"]), ("carry_cproj", w["w_proj"]), ("carry_fc1", w["w_fc2"]),
                                     ("carry_fc2", blocks[i + 1]["w_q"] if i + 1 < self.depth else None)):
                        cp, cn = rng(t) if t is not None else (None, 0)
                        setattr(d, cname, cp)
                        setattr(d, cname + "_bytes", cn)
                fd["desc"], fd["desc_key"] = arr, dkey
            f = _lib.DitForwardFold(
                dtype=ops.dtype_code(dt), Be=Be, N=N, D=D, H=H, dh=dh, hidden=hid.shape[1], depth=self.depth, L=L, nq_pad=nq_pad,
                nkv_pad_c=Kc.shape[2], nkv_pad_b=Kn_blk[0].shape[2] if dedup else 0, b_from=B if dedup else Be, step=prow,
                n_steps=plan["t"].numel(), ln_eps=self.LN_EPS, scale=scale, h=h.data_ptr(), xn=xn.data_ptr(), att=att.data_ptr(),
                hid=hid.data_ptr(), Qc=Qc.data_ptr(), Qs=Qs.data_ptr(), Ks=Ks.data_ptr(), Vs=Vs.data_ptr(), mod=mod.data_ptr(),
                center0=fcent[0].data_ptr(), center1=fcent[1].data_ptr(), part=fpart.data_ptr())
            if ride:
                f.kv_A, f.kv_W, f.kv_bias = y16.data_ptr(), pk["w_kv_all"].data_ptr(), pk["b_kv_all"].data_ptr()
                f.kv_rows, f.kv_rows_per_batch, f.kv_K = Bkv * Lk, Lk, Dc
            ops._dev(h, "h", torch.float32)                 # (the launch-device check of every op: ops._stream)
            _lib.check(_lib.load().primx_dit_blocks_fold(C.byref(f), fd["desc"], ops._stream()), "primx_dit_blocks_fold")
        else:
            for i, w in enumerate(pk["blocks"]):
                block(i, w, 0, Be)
                if self.block_probe is not None:
                    rs = h.std(-1)
                    self.block_probe.append({
                        "block": i, "folded": fold_uv is not None, "dtype": str(dt).replace("torch.", ""),
                        "residual_abs_max": float(h.abs().max()), "row_std_min": float(rs.min()), "row_std_median": float(rs.median()),
                        "row_std_max": float(rs.max()), "row_mean_abs_max": float(h.mean(-1).abs().max()),
                        "next_operand_abs_max": float(xn.float().abs().max()), "next_operand_finite": bool(torch.isfinite(xn.float()).all())})

        # ---- final layer (dit_crossattn.py:74-78)
        if not (fuse and self.depth):                    # (fused: the last block's fc2 launch has normalised the rows)
            ops.layernorm_modulate(h, fin_mod[0], fin_mod[1], N, xn, self.LN_EPS)
        out = ops.linear(xn, pk["w_final"], pk["b_final"])
        return out.view(Be, N, self.out_channels)

    def _forward_fp32(self, x, t, y):
        """The reference with autocast off: every Linear, the attention core, LayerNorm, modulate, GELU and the gated
        residuals in fp32 (dit_crossattn.py:184-202 with enable_amp=False).  Reads the fp32 parameters directly."""
        if self._packed_only:
            raise RuntimeError("this module holds only packed 16-bit weights (sharding.broadcast_packed_ / pack_from_state_dict "
                               "/ load_packed): the fp32 route needs the fp32 parameters - use broadcast_module_ / "
                               "load_state_dict")
        Be, N, Cin = x.shape
        L, Dc = y.shape[1], y.shape[2]
        D, H = self.hidden_size, self.num_heads
        dh = D // H
        T = Be * N
        dev = x.device
        f = lambda p: p.detach()
        h = torch.empty(T, D, dtype=torch.float32, device=dev)
        self._embed_tokens(x.reshape(T, Cin).float().contiguous(), h)
        st = ops.silu_f32(self.t_embedder(t))                                      # SiLU(t_emb), shared by every adaLN
        y2 = y.float().contiguous().view(Be * L, Dc)
        scale = dh ** -0.5
        xn = torch.empty(T, D, dtype=torch.float32, device=dev)
        for blk in self.blocks:
            ada = blk.adaLN_modulation[1]
            mod = ops.gemm_f32(st, f(ada.weight), f(ada.bias))                     # [Be, 9D]
            ch = [mod[:, j * D:(j + 1) * D] for j in range(9)]
            ca, sa, mlp = blk.crossattn, blk.attn, blk.mlp
            bias = lambda lin: None if lin.bias is None else f(lin.bias)
            # ---- cross-attention: q = scale * to_q(x) and the core scales by dh^-1/2 again (attention.py:105,109)
            ops.layernorm_modulate_f32(h, ch[0], ch[1], N, self.LN_EPS, out=xn)
            q = ops.gemm_f32(xn, f(ca.to_q.weight), bias(ca.to_q), out_scale=scale)
            k = ops.gemm_f32(y2, f(ca.to_k.weight), bias(ca.to_k))
            v = ops.gemm_f32(y2, f(ca.to_v.weight), bias(ca.to_v))
            o = ops.attention_f32(q.view(Be, N, H, dh), k.view(Be, L, H, dh), v.view(Be, L, H, dh), scale)
            ops.gemm_f32(o.view(T, D), f(ca.proj.weight), bias(ca.proj), out=h, gate=ch[2], rows_per_batch=N)
            # ---- self-attention on the unbind() views of the fused qkv buffer (attention.py:49-54)
            ops.layernorm_modulate_f32(h, ch[3], ch[4], N, self.LN_EPS, out=xn)
            qkv = ops.gemm_f32(xn, f(sa.qkv.weight), bias(sa.qkv)).view(Be, N, 3, H, dh)
            o = ops.attention_f32(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], scale)
            ops.gemm_f32(o.view(T, D), f(sa.proj.weight), bias(sa.proj), out=h, gate=ch[5], rows_per_batch=N)
            # ---- MLP
            ops.layernorm_modulate_f32(h, ch[6], ch[7], N, self.LN_EPS, out=xn)
            hid = ops.gemm_f32(xn, f(mlp.fc1.weight), f(mlp.fc1.bias), act=ACT_GELU_TANH)
            ops.gemm_f32(hid, f(mlp.fc2.weight), f(mlp.fc2.bias), out=h, gate=ch[8], rows_per_batch=N)
        fl = self.final_layer
        mod = ops.gemm_f32(st, f(fl.adaLN_modulation[1].weight), f(fl.adaLN_modulation[1].bias))
        ops.layernorm_modulate_f32(h, mod[:, :D], mod[:, D:2 * D], N, self.LN_EPS, out=xn)
        out = ops.gemm_f32(xn, f(fl.linear.weight), f(fl.linear.bias))
        return out.view(Be, N, self.out_channels)

    def _embed_tokens(self, xf: torch.Tensor, out: torch.Tensor, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[T, C] fp32 -> out [T, D] fp32 (and the same rows into `out2`), outside autocast in the reference
        (dit_crossattn.py:191-192)."""
        if out2 is not None and xf.shape[0] <= 8:      # (the few-row kernel has one destination)
            ops.linear_f32(xf, self.x_embedder.weight.detach(), self.x_embedder.bias.detach(), out=out)
            out2.copy_(out)
            return out
        return ops.linear_f32(xf, self.x_embedder.weight.detach(), self.x_embedder.bias.detach(), out=out, out2=out2)

    @ops.on_input_device
    def forward_with_cfg(self, x, t, y, cfg_scale=0.0, precision_dtype=torch.float32, enable_amp=False):
        """Classifier-free guidance: one forward at 2B, combine on all channels, return the B-sized half
        (dit_crossattn.py:204-213)."""
        self._check_inputs(x)
        if enable_amp and precision_dtype in (torch.float16, torch.bfloat16):
            model_out = self._forward16(x, t, y, precision_dtype, null_half=True)
        else:   # fp32 route: the reference's literal concatenations (dit_crossattn.py:205-208)
            y_null = self.null_cond_embedding.detach().to(y.dtype).expand_as(y)
            model_out = self.forward(torch.cat([x, x], dim=0), torch.cat([t, t], dim=0), torch.cat([y, y_null], dim=0),
                                     precision_dtype, enable_amp)
        return ops.cfg_combine(model_out.contiguous(), float(cfg_scale))


class PointEmbed(nn.Module):
    """Fourier features of the primitive position -> Linear (models/dit_crossattn.py:80-108).  Parameter container with
    the reference's buffer (`basis`, 3 x hidden_dim/2, block-diagonal 2^k pi) and `mlp` Linear(hidden_dim + 3, dim)."""

    def __init__(self, hidden_dim=48, dim=128):
        super().__init__()
        assert hidden_dim % 6 == 0
        self.embedding_dim = hidden_dim
        n = hidden_dim // 6
        e = torch.pow(2, torch.arange(n)).float() * math.pi
        z = torch.zeros(n)
        self.register_buffer("basis", torch.stack([torch.cat([e, z, z]), torch.cat([z, e, z]), torch.cat([z, z, e])]))
        self.mlp = nn.Linear(hidden_dim + 3, dim)


class DiTAdditivePosEmb(DiT):
    """The reference's second DiT class (models/dit_crossattn.py:215-301): token embedding = x_embedder(x) +
    point_emb(x[:, :, 1:4]); no condition dropout, hence no `null_cond_embedding` and no `forward_with_cfg`."""

    def __init__(self, seq_length=2, in_channels=4, condition_channels=512, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, attn_proj_bias=False, learn_sigma=True, gradient_checkpointing=False):
        super().__init__(seq_length=seq_length, in_channels=in_channels, condition_channels=condition_channels,
                         hidden_size=hidden_size, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                         cond_drop_prob=0.0, attn_proj_bias=attn_proj_bias, learn_sigma=learn_sigma,
                         gradient_checkpointing=gradient_checkpointing)
        self.point_emb = PointEmbed(hidden_dim=48, dim=hidden_size)

    def _embed_tokens(self, xf: torch.Tensor, out: torch.Tensor, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
        pe = self.point_emb
        n = pe.embedding_dim // 6
        feat = ops.point_features(xf, pe.basis[0, :n].contiguous())       # the non-zero entries of the block-diagonal basis
        w = pe.mlp.weight.detach()
        w = torch.nn.functional.pad(w, (0, feat.shape[1] - w.shape[1]))   # K padded like the features (zero column)
        out.copy_(ops.linear_f32(xf, self.x_embedder.weight.detach(), self.x_embedder.bias.detach()) +
                  ops.linear_f32(feat, w.contiguous(), pe.mlp.bias.detach()))
        if out2 is not None:
            out2.copy_(out)
        return out

    def _small_fp32_params(self):
        # the position-embedding Linear runs in fp32 next to x_embedder: it travels with the packed routes too
        return super()._small_fp32_params() + [self.point_emb.mlp.weight, self.point_emb.mlp.bias]

    def forward_with_cfg(self, *args, **kwargs):
        raise AttributeError("the reference's DiTAdditivePosEmb defines no forward_with_cfg (dit_crossattn.py:215-301)")
