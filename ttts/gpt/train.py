"""ttts/gpt/train.py -> ttts_amd.gpt.train: `Trainer(cfg_path='ttts/gpt/config.json').train()`, `save`, `load`
(reference :41-145).  `python -m ttts.gpt.train [config.json]` is the reference's `python ttts/gpt/train.py`."""
import sys

from ttts_amd.gpt.train import SyntheticGptBatches, Trainer, clean_checkpoints, cycle, get_grad_norm, warmup  # noqa: F401

if __name__ == "__main__":
    trainer = Trainer(*sys.argv[1:2])
    trainer.train()
