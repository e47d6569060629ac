"""ttts/gpt/model.py -> ttts_amd.gpt.model (UnifiedVoice: same constructor / forward / state-dict surface)."""
from ttts_amd.gpt.model import FusedAdamW, UnifiedVoice, prepare_tokens  # noqa: F401
