"""ttts/vqvae/augment -> ttts_amd.vqvae.augment (parametric-equaliser stage; Praat stays outside)."""
from ttts_amd.vqvae.augment import Augment, ParametricEqualizer  # noqa: F401
