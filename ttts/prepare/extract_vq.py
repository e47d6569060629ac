"""ttts/prepare/extract_vq.py -> ttts_amd.prepare.extract_vq."""
from ttts_amd.prepare.extract_vq import extract_vq_codes, process_vq, save_vq  # noqa: F401
