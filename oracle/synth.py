"""Deterministic synthetic weights / inputs shared by the golden-vector generator, the tests and bench.py.

Independent of torch's RNG and of module construction order: every tensor is drawn from a numpy
PCG64 stream keyed by (seed, crc32(name)), so the real reference (in make_golden.py), the oracle and
the HIP modules all load byte-identical values from nothing but shapes and names.

"Vacuous-parity trap" (SURVEY.md section 7): the reference zero-initialises every adaLN Linear and
the final Linear (models/dit_crossattn.py:173-182), which makes a fresh DiT output exactly 0.  All
tensors here are non-zero.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def tensor(seed: int, name: str, shape: Iterable[int], std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    a = _rng(seed, name).standard_normal(tuple(shape), dtype=np.float32) * np.float32(std) + np.float32(mean)
    return torch.from_numpy(a.astype(np.float32))


def dit_shapes(in_channels: int, condition_channels: int, hidden_size: int, depth: int, mlp_ratio: float = 4.0,
               learn_sigma: bool = True) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape of models.dit_crossattn.DiT (cond_drop_prob > 0, attn_proj_bias=True)."""
    D, Dc = hidden_size, condition_channels
    Hm = int(D * mlp_ratio)
    out_c = in_channels * 2 if learn_sigma else in_channels
    s: Dict[str, Tuple[int, ...]] = {
        "null_cond_embedding": (Dc,),
        "x_embedder.weight": (D, in_channels), "x_embedder.bias": (D,),
        "t_embedder.mlp.0.weight": (D, 256), "t_embedder.mlp.0.bias": (D,),
        "t_embedder.mlp.2.weight": (D, D), "t_embedder.mlp.2.bias": (D,),
        "final_layer.linear.weight": (out_c, D), "final_layer.linear.bias": (out_c,),
        "final_layer.adaLN_modulation.1.weight": (2 * D, D), "final_layer.adaLN_modulation.1.bias": (2 * D,),
    }
    for i in range(depth):
        p = f"blocks.{i}."
        s.update({
            p + "crossattn.to_q.weight": (D, D), p + "crossattn.to_q.bias": (D,),
            p + "crossattn.to_k.weight": (D, Dc), p + "crossattn.to_k.bias": (D,),
            p + "crossattn.to_v.weight": (D, Dc), p + "crossattn.to_v.bias": (D,),
            p + "crossattn.proj.weight": (D, D), p + "crossattn.proj.bias": (D,),
            p + "attn.qkv.weight": (3 * D, D), p + "attn.qkv.bias": (3 * D,),
            p + "attn.proj.weight": (D, D), p + "attn.proj.bias": (D,),
            p + "mlp.fc1.weight": (Hm, D), p + "mlp.fc1.bias": (Hm,),
            p + "mlp.fc2.weight": (D, Hm), p + "mlp.fc2.bias": (D,),
            p + "adaLN_modulation.1.weight": (9 * D, D), p + "adaLN_modulation.1.bias": (9 * D,),
        })
    return s


def dit_state_dict(seed: int, **cfg) -> Dict[str, torch.Tensor]:
    """Trained-network-like magnitudes: Linear weights ~ N(0, 1/fan_in) (so activations stay O(1)
    through depth), biases and adaLN rows ~ N(0, 0.02..0.05): gates are small but non-zero."""
    sd = {}
    for name, shape in dit_shapes(**cfg).items():
        if name == "null_cond_embedding":
            sd[name] = tensor(seed, name, shape, 1.0)
        elif name.endswith("bias"):
            sd[name] = tensor(seed, name, shape, 0.05)
        elif "adaLN_modulation" in name:
            sd[name] = tensor(seed, name, shape, 0.6 / shape[1] ** 0.5)
        else:
            sd[name] = tensor(seed, name, shape, 1.0 / shape[1] ** 0.5)
    return sd


def state_dict_like(seed: int, reference_sd: Dict[str, torch.Tensor], gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Synthetic values for an arbitrary module's state_dict (used for the VAE, whose key set is
    taken from the module itself): conv/linear weights ~ N(0, gain/fan_in), norm weights ~ 1 + N(0, .1),
    biases ~ N(0, .05)."""
    out = {}
    for name, ref in reference_sd.items():
        shape = tuple(ref.shape)
        if name.endswith("bias"):
            out[name] = tensor(seed, name, shape, 0.05)
        elif ".norm" in name or name.startswith("norm") or "norm_out" in name or ".norm." in name:
            out[name] = tensor(seed, name, shape, 0.1, 1.0)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            out[name] = tensor(seed, name, shape, (gain / max(fan_in, 1)) ** 0.5)
    return out
