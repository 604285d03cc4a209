"""models.dit_crossattn of the reference -> the HIP path (same class names, constructor kwargs, state_dict keys)."""
from topia_xl_amd.dit import DiT, DiTAdditivePosEmb, DiTBlock, FinalLayer, PointEmbed  # noqa: F401
from topia_xl_amd.dit import Mlp, TimestepEmbedder, modulate  # noqa: F401  (re-exported by the reference module too)
from topia_xl_amd.attention import MemEffAttention, MemEffCrossAttention  # noqa: F401

__primx_override__ = True
