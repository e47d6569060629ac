"""ttts/utils/vc_utils.py -> ttts_amd.utils.vc_utils."""
from ttts_amd.utils.vc_utils import HParams, get_logger, latest_checkpoint_path, load_checkpoint, save_checkpoint  # noqa: F401
