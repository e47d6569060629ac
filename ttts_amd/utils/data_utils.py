"""Mel / STFT front-end with the reference's function surface (ttts/utils/data_utils.py:21-27,52-156,158-187) on the
HIP kernels (`ttts_stft_mag_*`, `ttts_mel_log_*`): `spectrogram_torch`, `spec_to_mel_torch`, `mel_spectrogram_torch`
(all differentiable w.r.t. the waveform / spectrogram), `HParams`.  GPU tensors only -- no torch.stft fallback.

`librosa.filters.mel` (Slaney mel scale, Slaney area normalisation -- the reference's filterbank, data_utils.py:15,
95-97) is restated in `slaney_mel_basis`; librosa itself is not a dependency.
"""
import numpy as np
import torch

from .. import ops

mel_basis = {}
hann_window = {}


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """(n_mels, n_fft//2 + 1) float32: triangular filters on the Slaney mel scale, each normalised to unit area."""
    fmax = sr / 2.0 if fmax is None else fmax
    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    hz = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    d = np.diff(hz)
    ramps = hz[:, None] - fft_f[None, :]
    w = np.maximum(0.0, np.minimum(-ramps[:-2] / d[:-1, None], ramps[2:] / d[1:, None]))
    w *= (2.0 / (hz[2:n_mels + 2] - hz[:n_mels]))[:, None]
    return w.astype(np.float32)


def dynamic_range_compression_torch(x, C=1, clip_val=1e-5):
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression_torch(x, C=1):
    return torch.exp(x) / C


class _StftMag(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, window, n_fft, hop):
        ctx.save_for_backward(y, window)
        ctx.cfg = (n_fft, hop)
        return ops.stft_mag(y, window, n_fft, hop)

    @staticmethod
    def backward(ctx, dspec):
        y, window = ctx.saved_tensors
        n_fft, hop = ctx.cfg
        return ops.stft_mag_bwd(y, window, dspec.float(), n_fft, hop), None, None, None


class _MelLog(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, basis):
        mel = ops.mel_log(spec, basis)
        ctx.save_for_backward(mel, basis)
        ctx.n_bins = spec.shape[1]
        return mel

    @staticmethod
    def backward(ctx, dmel):
        mel, basis = ctx.saved_tensors
        return ops.mel_log_bwd(dmel.float(), mel, basis, ctx.n_bins), None


def _window(win_size, y):
    key = str(win_size) + "_" + str(y.dtype) + "_" + str(y.device)
    if key not in hann_window:
        hann_window[key] = torch.hann_window(win_size).to(dtype=y.dtype, device=y.device)
    return hann_window[key]


def _basis(spec_like, n_fft, num_mels, sampling_rate, fmin, fmax):
    key = "_".join(str(v) for v in (fmax, fmin, num_mels, n_fft, sampling_rate, spec_like.dtype, spec_like.device))
    if key not in mel_basis:
        mel_basis[key] = torch.from_numpy(slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax)).to(
            dtype=spec_like.dtype, device=spec_like.device)
    return mel_basis[key]


def spectrogram_torch(y, n_fft, hop_size, win_size, center=False):
    """(B, T) fp32 -> (B, n_fft//2 + 1, frames): reflect pad (n_fft - hop)/2, hann, |STFT| with the 1e-6 floor."""
    if center or win_size != n_fft:
        raise NotImplementedError("the training path uses center=False and win_size == n_fft (vqvae/config.json:57-60)")
    return _StftMag.apply(y, _window(win_size, y), n_fft, hop_size)


def spec_to_mel_torch(spec, n_fft, num_mels, sampling_rate, fmin, fmax):
    return _MelLog.apply(spec, _basis(spec, n_fft, num_mels, sampling_rate, fmin, fmax))


def mel_spectrogram_torch(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    return spec_to_mel_torch(spectrogram_torch(y, n_fft, hop_size, win_size, center), n_fft, num_mels, sampling_rate,
                             fmin, fmax)


class HParams:
    """Nested attribute view of a JSON config (ttts/utils/data_utils.py:158-187)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            if type(v) == dict:
                v = HParams(**v)
            self[k] = v

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def __len__(self):
        return len(self.__dict__)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        return setattr(self, key, value)

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        return self.__dict__.__repr__()
