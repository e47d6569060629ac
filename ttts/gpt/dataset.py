"""ttts/gpt/dataset.py -> ttts_amd.gpt.dataset."""
from ttts_amd.gpt.dataset import GptTtsCollater, GptTtsDataset  # noqa: F401
