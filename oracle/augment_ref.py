"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's parametric-equaliser augmentation.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Restates (reference = /root/reference, adelacvg/ttts), in float64 numpy:
* `ParametricEqualizer.biquad / low_shelving / high_shelving / peaking_equalizer`   ttts/vqvae/augment/peq.py:19-116
* `Augment.__init__` peak centres and `Augment.forward` without the Praat stage      ttts/vqvae/augment/__init__.py:23-97
  (torch.stft(center=True, hann, onesided) -> filters -> torch.istft -> clamp(-1, 1) -> peak normalisation)

Parity pin: `tests/golden/vqvae_peq.npz`, produced by `tools/make_goldens.py peq` from the imported reference
(`parselmouth`, which the reference imports for the excluded Praat stage, is stubbed there).
"""
import numpy as np


def biquad(a, b, windows):
    """peq.py:19-30: fir / iir on the rfft grid; a, b [..., 3]."""
    return np.fft.rfft(np.asarray(b, np.float64), windows, axis=-1) / np.fft.rfft(np.asarray(a, np.float64), windows, axis=-1)


def low_shelving(cutoff, gain, q, sr, windows):
    """peq.py:32-62."""
    gain, q = np.asarray(gain, np.float64), np.asarray(q, np.float64)
    w0 = 2 * np.pi * cutoff / sr
    alpha = np.sin(w0) / 2 / q
    c = np.cos(w0)
    A = np.exp(gain / 40. * np.log(10))
    b0 = A * ((A + 1) - (A - 1) * c + 2 * np.sqrt(A) * alpha)
    b1 = 2 * A * ((A - 1) - (A + 1) * c)
    b2 = A * ((A + 1) - (A - 1) * c - 2 * np.sqrt(A) * alpha)
    a0 = (A + 1) + (A - 1) * c + 2 * np.sqrt(A) * alpha
    a1 = -2 * ((A - 1) + (A + 1) * c)
    a2 = (A + 1) + (A - 1) * c - 2 * np.sqrt(A) * alpha
    return biquad(np.stack([a0, a1, a2], -1), np.stack([b0, b1, b2], -1), windows)


def high_shelving(cutoff, gain, q, sr, windows):
    """peq.py:64-95."""
    gain, q = np.asarray(gain, np.float64), np.asarray(q, np.float64)
    w0 = 2 * np.pi * cutoff / sr
    alpha = np.sin(w0) / 2 / q
    c = np.cos(w0)
    A = np.exp(gain / 40. * np.log(10))
    b0 = A * ((A + 1) + (A - 1) * c + 2 * np.sqrt(A) * alpha)
    b1 = -2 * A * ((A - 1) + (A + 1) * c)
    b2 = A * ((A + 1) + (A - 1) * c - 2 * np.sqrt(A) * alpha)
    a0 = (A + 1) - (A - 1) * c + 2 * np.sqrt(A) * alpha
    a1 = 2 * ((A - 1) - (A + 1) * c)
    a2 = (A + 1) - (A - 1) * c - 2 * np.sqrt(A) * alpha
    return biquad(np.stack([a0, a1, a2], -1), np.stack([b0, b1, b2], -1), windows)


def peaking_equalizer(center, gain, q, sr, windows):
    """peq.py:97-116."""
    center, gain, q = (np.asarray(t, np.float64) for t in (center, gain, q))
    w0 = 2 * np.pi * center / sr
    alpha = np.sin(w0) / 2 / q
    c = np.cos(w0)
    A = np.exp(gain / 40. * np.log(10))
    return biquad(np.stack([1 + alpha / A, -2 * c, 1 - alpha / A], -1), np.stack([1 + alpha * A, -2 * c, 1 - alpha * A], -1),
                  windows)


def peak_centers(f_min, f_max, peaks):
    """augment/__init__.py:31-36."""
    return f_min * (f_max / f_min) ** (np.arange(peaks + 2)[1:-1] / (peaks + 1))


def filters(quality_power, gain, cfg):
    """augment/__init__.py:65-86; cfg: dict(sampling_rate, win_length, cutoff_lowpass, cutoff_highpass, num_peak, q_min, q_max)."""
    quality_power, gain = np.asarray(quality_power, np.float64), np.asarray(gain, np.float64)
    q = cfg["q_min"] * (cfg["q_max"] / cfg["q_min"]) ** quality_power
    sr, win = cfg["sampling_rate"], cfg["win_length"]
    center = np.broadcast_to(peak_centers(cfg["cutoff_lowpass"], cfg["cutoff_highpass"], cfg["num_peak"])[None],
                             q[:, :-2].shape)
    peaks = np.prod(peaking_equalizer(center, gain[:, :-2], q[:, :-2], sr, win), axis=1)
    low = low_shelving(cfg["cutoff_lowpass"], gain[:, -2], q[:, -2], sr, win)
    high = high_shelving(cfg["cutoff_highpass"], gain[:, -1], q[:, -1], sr, win)
    return peaks * high * low


def hann(n):
    """torch.hann_window(n) (periodic)."""
    return 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)


def stft_center(wav, n_fft, hop):
    """torch.stft(wav, n_fft, hop, n_fft, hann, center=True, pad_mode='reflect', return_complex=True): [B, F, frames]."""
    wav = np.asarray(wav, np.float64)
    x = np.pad(wav, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    frames = 1 + wav.shape[1] // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
    return np.fft.rfft(x[:, idx] * hann(n_fft), axis=-1).transpose(0, 2, 1)


def istft_center(spec, n_fft, hop):
    """torch.istft(spec, n_fft, hop, n_fft, hann) (center=True, length=None): [B, hop * (frames - 1)]."""
    B, _, frames = spec.shape
    w = hann(n_fft)
    fr = np.fft.irfft(spec.transpose(0, 2, 1), n_fft, axis=-1) * w
    total = n_fft + hop * (frames - 1)
    y = np.zeros((B, total))
    env = np.zeros(total)
    for t in range(frames):
        y[:, t * hop:t * hop + n_fft] += fr[:, t]
        env[t * hop:t * hop + n_fft] += w * w
    s = n_fft // 2
    return y[:, s:s + hop * (frames - 1)] / env[s:s + hop * (frames - 1)]


def augment_forward(wav, quality_power, gain, cfg):
    """Augment.forward with pitch_shift = pitch_range = formant_shift = None (augment/__init__.py:57-97)."""
    n_fft, hop = cfg["win_length"], cfg["hop_length"]
    fft = stft_center(wav, n_fft, hop)
    if quality_power is not None:
        fft = fft * filters(quality_power, gain, cfg)[..., None]
    out = np.clip(istft_center(fft, n_fft, hop), -1., 1.)
    return out / np.maximum(np.abs(out).max(axis=-1, keepdims=True), 1e-7)
