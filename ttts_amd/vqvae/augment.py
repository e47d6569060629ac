"""Waveform augmentation of the VQ-VAE step: the parametric-equaliser path of the reference's `Augment`
(ttts/vqvae/augment/__init__.py:11-116, augment/peq.py:6-116) on the HIP kernels of csrc/peq.hip, plus the
parameter sampler and NaN-retry loop of the trainer (ttts/vqvae/train.py:62-116).

Same names, arguments and tensor shapes as the reference.  Not built: the Praat stage (`PraatAugment`,
augment/praat.py:6-57 -- formant / pitch shifting through the parselmouth CPU library).  `Augment.forward` raises
NotImplementedError when one of `pitch_shift / pitch_range / formant_shift` is given, and `augment()` therefore never
passes them (the reference always passes formant_shift and pitch_range, i.e. always runs Praat).
"""
import torch
import torch.nn as nn

from .. import ops


class ParametricEqualizer(nn.Module):
    """Frequency responses of biquad IIR filters on the `windows`-point rfft grid (peq.py:6-116).
    Each method returns complex64 [..., windows // 2 + 1] like the reference (fir / iir of two 3-tap rffts)."""

    def __init__(self, sr, windows):
        super().__init__()
        self.sr = sr
        self.windows = windows

    def _one(self, kind, freq, gain, q):
        gain = torch.as_tensor(gain, dtype=torch.float32)
        shape = gain.shape
        dev = gain.device
        freq = torch.as_tensor(freq, dtype=torch.float32, device=dev).expand(shape)
        q = torch.as_tensor(q, dtype=torch.float32, device=dev).expand(shape)
        k = torch.tensor([kind], dtype=torch.int32, device=dev)
        H = ops.peq_response(freq.reshape(-1, 1).contiguous(), gain.reshape(-1, 1).contiguous(), q.reshape(-1, 1).contiguous(),
                             k, self.windows, self.sr)
        return H.reshape(*shape, self.windows // 2 + 1)

    def low_shelving(self, cutoff, gain, q):
        return self._one(ops.PEQ_LOW_SHELF, cutoff, gain, q)

    def high_shelving(self, cutoff, gain, q):
        return self._one(ops.PEQ_HIGH_SHELF, cutoff, gain, q)

    def peaking_equalizer(self, center, gain, q):
        return self._one(ops.PEQ_PEAK, center, gain, q)


class Augment(nn.Module):
    """Waveform augmentation (augment/__init__.py:11-116).  `config` is the HParams tree of vqvae/config.json
    (data.sampling_rate / win_length / hop_length, train.cutoff_lowpass / cutoff_highpass / num_peak / q_min / q_max)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.praat = None
        self.peq = ParametricEqualizer(config.data.sampling_rate, config.data.win_length)
        self.register_buffer("window", torch.hann_window(config.data.win_length), persistent=False)
        f_min, f_max, peaks = config.train.cutoff_lowpass, config.train.cutoff_highpass, config.train.num_peak
        self.register_buffer("peak_centers", f_min * (f_max / f_min) ** (torch.arange(peaks + 2)[1:-1] / (peaks + 1)),
                             persistent=False)

    def filters(self, quality_power, gain=None):
        """[B, F] complex64: product of num_peak peaking filters, the low shelf and the high shelf (:65-86), one launch."""
        c = self.config
        q = c.train.q_min * (c.train.q_max / c.train.q_min) ** quality_power.float()
        if gain is None:
            # the reference builds a [B, num_peak] zero tensor here and then slices it again, which cannot broadcast;
            # flat (0 dB) gains for all num_peak + 2 filters is what that branch means
            gain = torch.zeros_like(q)
        bsize, peaks = q.shape[0], q.shape[1] - 2
        dev = q.device
        freq = torch.cat([self.peak_centers.to(dev, torch.float32)[None].expand(bsize, peaks),
                          torch.full((bsize, 1), float(c.train.cutoff_lowpass), device=dev),
                          torch.full((bsize, 1), float(c.train.cutoff_highpass), device=dev)], dim=1).contiguous()
        kind = torch.tensor([ops.PEQ_PEAK] * peaks + [ops.PEQ_LOW_SHELF, ops.PEQ_HIGH_SHELF], dtype=torch.int32, device=dev)
        return ops.peq_response(freq, gain.float().contiguous(), q.contiguous(), kind, c.data.win_length, c.data.sampling_rate)

    @torch.no_grad()
    def forward(self, wavs, pitch_shift=None, pitch_range=None, formant_shift=None, quality_power=None, gain=None):
        if formant_shift is not None or pitch_shift is not None or pitch_range is not None:
            raise NotImplementedError("the Praat stage (formant / pitch shift) is a CPU library outside this build; "
                                      "pass quality_power / gain only")
        d = self.config.data
        H = self.filters(quality_power, gain) if quality_power is not None else None
        return ops.stft_filter_istft(wavs, self.window, d.win_length, d.hop_length, H, clamp=True, peak_normalize=True, eps=1e-7)


def sample_like(signal, hps, generator=None):
    """Augmentation parameters for a batch (train.py:62-88): (formant_shift, pitch_shift, pitch_range, power, gain)."""
    bsize = signal.shape[0]
    dev = signal.device
    tr = hps.train

    def sampler(ratio):
        shifts = torch.rand(bsize, device=dev, generator=generator) * (ratio - 1.) + 1.
        flip = (torch.rand(bsize, device=dev, generator=generator) < 0.5)
        return torch.where(flip, shifts ** -1, shifts)

    fs, ps, pr = sampler(tr.formant_shift), sampler(tr.pitch_shift), sampler(tr.pitch_range)
    peaks = tr.num_peak
    power = torch.rand(bsize, peaks + 2, device=dev, generator=generator)
    gain = torch.rand(bsize, peaks + 2, device=dev, generator=generator) * (tr.g_max - tr.g_min) + tr.g_min
    return fs, ps, pr, power, gain


def augment(signal, aug, hps, ps=False, generator=None):
    """Augment the speech (train.py:89-116): re-samples parameters until every clip has a NaN-free result.
    One host sync per call (the reference's `nan.all()`); the Praat parameters are sampled (same RNG stream) but unused."""
    bsize = signal.shape[0]
    saves = None
    while saves is None or len(saves) < bsize:
        _fshift, _pshift, _prange, power, gain = sample_like(signal, hps, generator)
        out = aug.forward(signal, quality_power=power, gain=gain)
        nan = out.isnan().any(dim=-1)
        if not bool(nan.any()):
            saves = out if saves is None else torch.cat([saves, out], dim=0)
        elif not bool(nan.all()):
            saves = out[~nan] if saves is None else torch.cat([saves, out[~nan]], dim=0)
    return saves[:bsize]
