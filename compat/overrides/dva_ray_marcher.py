"""dva.ray_marcher of the reference -> the HIP ray marcher (the CUDA extensions it imports at module top,
dva/ray_marcher.py:15-16, do not build on ROCm).  `generate_colored_boxes` is imported by dva/visualize.py:7."""
from topia_xl_amd.raymarch import RayMarcher, convert_camera_parameters, generate_colored_boxes  # noqa: F401

__primx_override__ = True
