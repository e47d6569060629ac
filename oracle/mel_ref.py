"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference mel/STFT front-end.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Restates (reference = /root/reference, adelacvg/ttts):
* `spectrogram_torch`               ttts/utils/data_utils.py:52-87   reflect-pad (n_fft-hop)/2, STFT
                                    (hann, center=False, onesided), sqrt(re^2 + im^2 + 1e-6)
* `spec_to_mel_torch`               ttts/utils/data_utils.py:90-103  mel-basis matmul, log(clamp(x, 1e-5))
* `mel_spectrogram_torch`           ttts/utils/data_utils.py:106-156
* `dynamic_range_compression_torch` ttts/utils/data_utils.py:21-27
* `librosa.filters.mel` (third-party, unpinned, absent from this image; call sites data_utils.py:15,95-97):
  the published Slaney-scale / Slaney-normalised triangular filterbank is restated in `slaney_mel_basis`.
  PARITY UNPINNED against a real librosa install; it is pinned against
  `transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney')` in `tools/make_goldens.py`.

Parity pin for the STFT / mel functions: fixtures `tests/golden/mel_*.npz` produced by importing the
reference (with the librosa stub above) in `tools/make_goldens.py`.
"""
import numpy as np
import torch


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3.0
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3.0
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """(n_mels, n_fft//2+1) float32 triangular filters, Slaney mel scale, area-normalised."""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels]))[:, None]
    return w.astype(np.float32)


def spectrogram(y, n_fft, hop_size, win_size, center=False):
    """y (B, T) -> (B, n_fft//2+1, frames)"""
    win = torch.hann_window(win_size).to(dtype=y.dtype, device=y.device)
    pad = int((n_fft - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=win, center=center,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.view_as_real(spec)
    return torch.sqrt(spec.pow(2).sum(-1) + 1e-6)


def spec_to_mel(spec, n_fft, num_mels, sampling_rate, fmin, fmax):
    basis = torch.from_numpy(slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax)).to(spec)
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    return spec_to_mel(spectrogram(y, n_fft, hop_size, win_size, center), n_fft, num_mels, sampling_rate, fmin, fmax)
