from topia_xl_amd.attention import MemEffAttention, MemEffCrossAttention  # noqa: F401
