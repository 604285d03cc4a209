from topia_xl_amd.dit import DiT, DiTAdditivePosEmb, DiTBlock, FinalLayer, PointEmbed  # noqa: F401
