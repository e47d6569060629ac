"""ttts/diffusion/train.py -> ttts_amd.diffusion.train."""
from ttts_amd.diffusion.train import DiffusionTrainer, warmup  # noqa: F401

Trainer = DiffusionTrainer
