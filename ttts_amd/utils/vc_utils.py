"""Checkpoint helpers of ttts/utils/vc_utils.py:248-330 (`save_checkpoint`, `load_checkpoint`,
`latest_checkpoint_path`): dict layout `{'model','iteration','optimizer','learning_rate'}`, tolerant load."""
from ..vqvae.train import latest_checkpoint_path, load_checkpoint, save_checkpoint  # noqa: F401
from .data_utils import HParams  # noqa: F401


def get_logger(model_dir, filename="train.log"):
    """vc_utils.py:380-394."""
    import logging
    import os
    logger = logging.getLogger(os.path.basename(model_dir))
    logger.setLevel(logging.DEBUG)
    os.makedirs(model_dir, exist_ok=True)
    if not logger.handlers:
        h = logging.FileHandler(os.path.join(model_dir, filename))
        h.setLevel(logging.DEBUG)
        h.setFormatter(logging.Formatter("%(asctime)s\t%(name)s\t%(levelname)s\t%(message)s"))
        logger.addHandler(h)
    return logger
