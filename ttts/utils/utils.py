"""ttts/utils/utils.py -> ttts_amd.utils.utils."""
from ttts_amd.utils.utils import clean_checkpoints  # noqa: F401
