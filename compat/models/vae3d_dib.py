from topia_xl_amd.vae import VAE, Decoder, Encoder, MidBlock, ResnetBlock, UpBlock, VolumeAttention  # noqa: F401
