"""models.attention of the reference -> the HIP attention modules (no xFormers import)."""
from topia_xl_amd.attention import MemEffAttention, MemEffCrossAttention  # noqa: F401
from topia_xl_amd.ops import memory_efficient_attention  # noqa: F401  (the name models/attention.py:17 binds)
from torch import unbind  # noqa: F401

__primx_override__ = True
