"""ttts/vqvae/quantize.py -> ttts_amd.vqvae.quantize."""
from ttts_amd.vqvae.quantize import ResidualVectorQuantizer  # noqa: F401
