"""ttts/utils/data_utils.py -> ttts_amd.utils.data_utils."""
from ttts_amd.utils.data_utils import (HParams, dynamic_range_compression_torch, dynamic_range_decompression_torch,  # noqa: F401
                                       mel_spectrogram_torch, spec_to_mel_torch, spectrogram_torch)
