from topia_xl_amd.primsdf import PrimSDF  # noqa: F401
