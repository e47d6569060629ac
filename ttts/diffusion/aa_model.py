"""ttts/diffusion/aa_model.py -> ttts_amd.diffusion.aa_model."""
from ttts_amd.diffusion.aa_model import (AA_diffusion, AttentionBlock, DiffusionLayer, RefEncoder, ResBlock,  # noqa: F401
                                         denormalize_tacotron_mel, normalize_tacotron_mel, timestep_embedding)
