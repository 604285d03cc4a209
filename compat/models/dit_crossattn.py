from topia_xl_amd.dit import DiT, DiTBlock, FinalLayer  # noqa: F401
