"""ttts/vqvae/core_vq.py -> ttts_amd.vqvae.quantize."""
from ttts_amd.vqvae.quantize import EuclideanCodebook, ResidualVectorQuantization, VectorQuantization  # noqa: F401
