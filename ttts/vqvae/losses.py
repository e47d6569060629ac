"""ttts/vqvae/losses.py -> ttts_amd.vqvae.losses."""
from ttts_amd.vqvae.losses import discriminator_loss, feature_loss, generator_loss, kl_loss  # noqa: F401
