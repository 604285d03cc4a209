"""Drop-in for the reference's ``models.diffusion`` package (models/diffusion/__init__.py:10-52)."""
from . import schedule as gaussian_diffusion_tables  # noqa: F401
from .sampler import GaussianDiffusion, SpacedDiffusion, create_diffusion  # noqa: F401
from .schedule import (LossType, ModelMeanType, ModelVarType, get_named_beta_schedule,  # noqa: F401
                       space_timesteps)
