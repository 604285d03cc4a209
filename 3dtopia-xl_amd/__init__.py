"""MI355X-native hot path of 3DTopia-XL: the DDIM denoising loop over the PrimX diffusion
transformer and the per-primitive 3D-VAE decode (import as ``topia_xl_amd``).

Public surface = the reference's own Python interface for this path:
``DiT`` (models/dit_crossattn.py), ``VAE`` (models/vae3d_dib.py), ``create_diffusion`` and the
samplers (models/diffusion), ``memory_efficient_attention`` (the xFormers seam), plus
``ShardedSampler`` for batch-sharded multi-GPU sampling.  All arithmetic runs in
``csrc/libprimx_hip.so`` (C ABI in include/primx_hip.h); importing the package does not need the
library, calling any op does.
"""
from .diffusion import create_diffusion, GaussianDiffusion, SpacedDiffusion, space_timesteps  # noqa: F401

__all__ = ["create_diffusion", "GaussianDiffusion", "SpacedDiffusion", "space_timesteps", "DiT", "DiTAdditivePosEmb", "VAE",
           "memory_efficient_attention"]


def __getattr__(name):  # lazy: torch.nn modules are only built when asked for
    if name == "DiT":
        from .dit import DiT
        return DiT
    if name == "DiTAdditivePosEmb":
        from .dit import DiTAdditivePosEmb
        return DiTAdditivePosEmb
    if name == "VAE":
        from .vae import VAE
        return VAE
    if name == "memory_efficient_attention":
        from .ops import memory_efficient_attention
        return memory_efficient_attention
    if name == "ShardedSampler":
        from .sharding import ShardedSampler
        return ShardedSampler
    raise AttributeError(name)
