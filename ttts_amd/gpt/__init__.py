from .engine import GptEngine, param_spec, resolve_config  # noqa: F401
from .model import FusedAdamW, UnifiedVoice, prepare_tokens  # noqa: F401
