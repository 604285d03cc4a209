"""models.primsdf of the reference -> the HIP field query (no trimesh import)."""
from topia_xl_amd.primsdf import PrimSDF  # noqa: F401

__primx_override__ = True
