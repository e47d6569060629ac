"""ttts/utils/diffusion.py -> ttts_amd.diffusion.gaussian (training_losses path)."""
from ttts_amd.diffusion.gaussian import SpacedDiffusion, get_named_beta_schedule, space_timesteps  # noqa: F401
