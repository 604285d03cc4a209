from topia_xl_amd.diffusion import *  # noqa: F401,F403
from topia_xl_amd.diffusion import create_diffusion, SpacedDiffusion, space_timesteps  # noqa: F401
