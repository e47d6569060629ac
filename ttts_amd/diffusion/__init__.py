"""Diffusion mel-denoiser step of the reference (ttts/diffusion/) on the HIP kernels."""
from .aa_model import AA_diffusion, denormalize_tacotron_mel, normalize_tacotron_mel, timestep_embedding  # noqa: F401
from .gaussian import SpacedDiffusion, get_named_beta_schedule, space_timesteps  # noqa: F401
