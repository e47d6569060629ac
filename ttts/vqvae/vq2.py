"""ttts/vqvae/vq2.py -> ttts_amd.vqvae.vq2."""
from ttts_amd.vqvae.modules import Generator  # noqa: F401
from ttts_amd.vqvae.vq2 import (MRTE, DiscriminatorP, DiscriminatorS, MultiPeriodDiscriminator, PosteriorAudioEncoder,  # noqa: F401
                                ResidualCouplingBlock, SynthesizerTrn, TextEncoder)
