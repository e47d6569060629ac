"""ttts/vqvae/modules.py -> ttts_amd.vqvae.modules / style_encoder."""
from ttts_amd.vqvae.attentions import LayerNorm  # noqa: F401
from ttts_amd.vqvae.modules import (WN, Activation1d, Flip, ResBlock1, ResidualCouplingLayer, SnakeBeta, get_padding)  # noqa: F401
from ttts_amd.vqvae.style_encoder import Conv1dGLU, ConvNorm, LinearNorm, MelStyleEncoder, Mish  # noqa: F401
