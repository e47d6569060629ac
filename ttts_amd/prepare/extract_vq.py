"""Offline VQ-code extraction: the step between the two training paths.

Mirrors ttts/prepare/extract_vq.py:8-22 (`process_vq`): run the trained VQ-VAE encoder + quantizer over an utterance
and store the code sequence as a plain Python list of ints with `torch.save(code.tolist(), path + '.vq.pth')` -- the
format `ttts/gpt/dataset.py:46-47` reads back with `LongTensor(torch.load(quant_path))`.

The reference calls `vqvae.extract_code(mel)` on a cached mel (a model generation older than vq2.SynthesizerTrn); for
the vq2 model the equivalent is `SynthesizerTrn.extract_latent(wav, spec)` (vq2.py:912-920), which is what runs here,
on the HIP kernels, under `torch.no_grad()` in eval mode (no EMA update, no dropout).
"""
import os

import torch

from ..utils.data_utils import spectrogram_torch


@torch.no_grad()
def extract_vq_codes(model, wav, hps_data, wav_lengths=None):
    """wav (B, T) fp32 on the GPU -> int64 codes (B, n_q, T_spec // 2)."""
    modes = [(m, m.training) for m in model.modules()]      # restore every sub-module's own flag (ref_enc may be in eval)
    model.eval()
    try:
        spec = spectrogram_torch(wav, hps_data.filter_length, hps_data.hop_length, hps_data.win_length, center=False)
        y_lengths = None if wav_lengths is None else torch.div(wav_lengths, hps_data.hop_length, rounding_mode="floor")
        return model.extract_latent(wav, spec, y_lengths)
    finally:
        for m, flag in modes:
            m.training = flag


def save_vq(path, codes_1d):
    """`path` is the audio path; writes `path + '.vq.pth'` holding a list of ints (reference on-disk format)."""
    outp = path + ".vq.pth"
    os.makedirs(os.path.dirname(outp) or ".", exist_ok=True)
    torch.save([int(v) for v in codes_1d.reshape(-1).tolist()], outp)
    return outp


def process_vq(model, path, wav, hps_data):
    """One utterance: wav (T,) or (1, T) tensor already at the model's sampling rate -> `<path>.vq.pth`."""
    wav = wav.reshape(1, -1).to(next(model.parameters()).device, torch.float32)
    codes = extract_vq_codes(model, wav, hps_data)
    return save_vq(path, codes[0, 0])
