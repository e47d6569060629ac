"""`clean_checkpoints` of ttts/utils/utils.py:67-85 (the GPT trainer's checkpoint rotation)."""
from ..gpt.train import clean_checkpoints  # noqa: F401
