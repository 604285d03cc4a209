"""models.diffusion of the reference (models/diffusion/__init__.py:10-52) -> the device-resident sampler.  Executed as
the package `models.diffusion` with the reference's own directory as search path, so `models.diffusion.respace`,
`.gaussian_diffusion`, ... still import from the reference for code that wants them."""
from topia_xl_amd.diffusion import *  # noqa: F401,F403
from topia_xl_amd.diffusion import create_diffusion, GaussianDiffusion, SpacedDiffusion, space_timesteps  # noqa: F401

__primx_override__ = True
