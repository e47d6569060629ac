"""ttts/utils/commons.py -> ttts_amd.utils.commons."""
from ttts_amd.utils.commons import clip_grad_value_, rand_slice_segments, sequence_mask, slice_segments  # noqa: F401
