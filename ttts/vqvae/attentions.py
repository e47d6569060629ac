"""ttts/vqvae/attentions.py -> ttts_amd.vqvae.attentions."""
from ttts_amd.vqvae.attentions import FFN, Encoder, LayerNorm, MultiHeadAttention  # noqa: F401
