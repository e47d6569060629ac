"""Import shim for the *reference* (`/root/reference`, adelacvg/ttts) -- golden-vector generation only.

This file only runs in the build container (where `/root/reference` is mounted read-only).
Nothing here is imported by the product (`ttts_amd/`), by `bench.py`, or by the `-m gpu` tests:
the reference cannot travel to the GPU box, only the fixtures it produces (`tests/golden/`) do.

The stubs replace third-party packages that are absent from this image and that the reference
imports at module import time without using them on the hot path (torchaudio, librosa utilities,
`transformers.utils.model_parallel_utils`).  `librosa.filters.mel` is the one stub that matters
numerically: it is bound to `transformers.audio_utils.mel_filter_bank(norm='slaney',
mel_scale='slaney')`, i.e. the Slaney filterbank librosa documents.  Parity of that filterbank
against a real librosa install is UNPINNED in this container.
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def install():
    sys.dont_write_bytecode = True  # reference tree is read-only; never emit .pyc there
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # the repo root carries its own regular package `ttts` (the drop-in import names, re-exporting ttts_amd); the reference's
    # `ttts` is a namespace package and would lose to it on any sys.path order -> bind the name to the reference explicitly
    for name in [m for m in sys.modules if m == "ttts" or m.startswith("ttts.")]:
        del sys.modules[name]
    ref_pkg = types.ModuleType("ttts")
    ref_pkg.__path__ = [REFERENCE_ROOT + "/ttts"]
    sys.modules["ttts"] = ref_pkg
    import torch  # noqa: F401
    import transformers  # noqa: F401  (real transformers first: its lazy module probes torchaudio)
    from transformers import GPT2Config, GPT2Model  # noqa: F401
    from transformers.audio_utils import mel_filter_bank

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    ta = stub("torchaudio")
    ta.__path__ = []
    stub("torchaudio.functional")
    stub("torchaudio.transforms")

    def librosa_mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **_):
        return mel_filter_bank(n_fft // 2 + 1, n_mels, fmin, fmax if fmax is not None else sr / 2, sr,
                               norm="slaney", mel_scale="slaney").T.astype(np.float32)

    lb = stub("librosa")
    lb.__path__ = []
    lb.util = stub("librosa.util", normalize=None, pad_center=None, tiny=None)
    lb.filters = stub("librosa.filters", mel=librosa_mel)
    stub("transformers.utils.model_parallel_utils", get_device_map=None, assert_device_map=None)
    kd = stub("k_diffusion")
    kd.__path__ = []
    stub("k_diffusion.sampling", sample_dpmpp_2m=None, sample_euler_ancestral=None)
    tf = sys.modules["transformers"]
    tf.LogitsWarper = tf.LogitsProcessor
    return librosa_mel
