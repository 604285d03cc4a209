"""models.vae3d_dib of the reference -> the HIP decoder (models/vae3d_dib.py:389-453 surface)."""
from topia_xl_amd.vae import VAE, Decoder, DownBlock, Encoder, MidBlock, ResnetBlock, UpBlock, VolumeAttention  # noqa: F401

__primx_override__ = True
