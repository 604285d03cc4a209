"""Import the UNMODIFIED reference modules from /root/reference (TEST INFRASTRUCTURE; works only in
the build container - the GPU box has no /root/reference, and nothing under tests -m gpu, smoke() or
bench.py calls this).

The reference needs ``xformers.ops`` (models/attention.py:17), a third-party package that is neither
vendored nor pinned (README.md:67 ``conda install xformers::xformers``) and is absent here.  A
stand-in with the documented semantics is registered in ``sys.modules`` before the import:
``memory_efficient_attention(q, k, v, attn_bias=None)`` on [B, M, H, K] = softmax(q k^T K^-0.5) v in
fp32 (float64 internally), ``unbind = torch.unbind``.  Nothing of the reference is edited or copied.
"""
from __future__ import annotations

import os
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


def _mea(q, k, v, attn_bias=None, p=0.0, scale=None):
    assert attn_bias is None and p == 0.0
    s = q.shape[-1] ** -0.5 if scale is None else scale
    qd, kd, vd = (z.double().permute(0, 2, 1, 3) for z in (q, k, v))
    out = torch.softmax(qd @ kd.transpose(-1, -2) * s, dim=-1) @ vd
    return out.permute(0, 2, 1, 3).to(q.dtype)


def load():
    """Returns (dit_module, vae_module, diffusion_package, attention_module) of the reference."""
    if not available():
        raise RuntimeError(f"{REFERENCE_ROOT} is not present on this machine")
    if "xformers" not in sys.modules:
        xf = types.ModuleType("xformers")
        xops = types.ModuleType("xformers.ops")
        xops.memory_efficient_attention = _mea
        xops.unbind = torch.unbind
        xf.ops = xops
        sys.modules["xformers"] = xf
        sys.modules["xformers.ops"] = xops
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    dit = importlib.import_module("models.dit_crossattn")
    vae = importlib.import_module("models.vae3d_dib")
    diffusion = importlib.import_module("models.diffusion")
    attention = importlib.import_module("models.attention")
    return dit, vae, diffusion, attention
