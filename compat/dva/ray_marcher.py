from topia_xl_amd.raymarch import RayMarcher, convert_camera_parameters  # noqa: F401
