"""VQ-VAE-GAN model classes on the HIP kernels, mirroring ttts/vqvae/vq2.py (constructor arguments, forward signatures,
return structure and state-dict keys).

Built so far: `Generator` (:341-415, in modules.py), `DiscriminatorP` (:418-494), `DiscriminatorS` (:497-524),
`MultiPeriodDiscriminator` (:527-551), `ResidualCouplingBlock` (:208-252), `PosteriorAudioEncoder` (:667-745),
`MRTE` (:17-46), `TextEncoder` (:90-164), `SynthesizerTrn` (:750-871, training forward).
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from . import attentions, modules
from .attentions import MultiHeadAttention
from .quantize import ResidualVectorQuantizer
from .style_encoder import MelStyleEncoder  # noqa: F401
from .modules import LRELU_SLOPE, Conv1d, Conv2dK1, Generator, get_padding  # noqa: F401


# SynthesizerTrn's two branches: the posterior / flow / decoder branch (index 1) runs on the caller's stream, only the prior branch on
# a side stream -- the decoder's own three-way fan-out then forks from the caller's stream too (siblings, not nested: see
# modules.run_branches).  TTTS_SYNTH_INLINE=-1 restores both branches on side streams (the nested form a capture cannot hold).
_SYNTH_INLINE = (lambda v: None if v < 0 else v)(int(os.environ.get("TTTS_SYNTH_INLINE", "1")))


def _no_spectral_norm(flag):
    if flag:
        raise NotImplementedError("use_spectral_norm=True is not on the training path (vqvae/config.json: false)")


class DiscriminatorP(nn.Module):
    """Period discriminator: the waveform folded to (T/p, p) and convolved along T/p with (5,1) kernels.  The period axis
    is folded into the batch, so every layer is a dense 1-D convolution on contiguous rows; feature maps are handed out
    as (B, C, H, W) views of that storage (no copies)."""

    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False):
        super().__init__()
        _no_spectral_norm(use_spectral_norm)
        self.period = period
        chans = [1, 32, 128, 512, 1024, 1024]
        self.convs = nn.ModuleList([
            Conv2dK1(chans[i], chans[i + 1], kernel_size, stride if i < 4 else 1,
                     padding=get_padding(kernel_size, 1)).apply_weight_norm("old") for i in range(5)])
        self.conv_post = Conv2dK1(1024, 1, 3, 1, padding=1).apply_weight_norm("old")

    def forward(self, x):
        fmap = []
        b, c, t = x.shape
        p = self.period
        if t % p != 0:
            n_pad = p - (t % p)
            x = F.pad(x, (0, n_pad), "reflect")
            t = t + n_pad
        x = x.view(b, t // p, p).permute(0, 2, 1).reshape(b * p, 1, t // p)        # (B*W, 1, H)

        def nchw(h):
            return h.view(b, p, h.shape[1], h.shape[2]).permute(0, 2, 3, 1)          # (B, C, H, W) view

        for l in self.convs:
            x = l(x, out_act="lrelu", out_slope=LRELU_SLOPE)
            fmap.append(nchw(x))
        x = self.conv_post(x)
        fmap.append(nchw(x))
        return torch.flatten(fmap[-1], 1, -1), fmap


class DiscriminatorS(nn.Module):
    def __init__(self, use_spectral_norm=False):
        super().__init__()
        _no_spectral_norm(use_spectral_norm)
        spec = [(1, 16, 15, 1, 1, 7), (16, 64, 41, 4, 4, 20), (64, 256, 41, 4, 16, 20), (256, 1024, 41, 4, 64, 20),
                (1024, 1024, 41, 4, 256, 20), (1024, 1024, 5, 1, 1, 2)]
        self.convs = nn.ModuleList([Conv1d(ci, co, k, s, padding=pd, groups=g).apply_weight_norm("old")
                                    for ci, co, k, s, g, pd in spec])
        self.conv_post = Conv1d(1024, 1, 3, 1, padding=1).apply_weight_norm("old")

    def forward(self, x):
        fmap = []
        for l in self.convs:
            x = l(x, out_act="lrelu", out_slope=LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class MultiPeriodDiscriminator(nn.Module):
    def __init__(self, use_spectral_norm=False):
        super().__init__()
        periods = [2, 3, 5, 7, 11]
        discs = [DiscriminatorS(use_spectral_norm=use_spectral_norm)]
        discs = discs + [DiscriminatorP(i, use_spectral_norm=use_spectral_norm) for i in periods]
        self.discriminators = nn.ModuleList(discs)

    def forward(self, y, y_hat):
        # The six sub-discriminators are independent, and most of their launches either under-fill the chip or end in a nearly
        # empty last round of workgroups (DiscriminatorP's 1024-channel layers: 528..608 workgroups of 64 x 256 outputs on 512
        # slots): spread over TTTS_D_STREAMS (default 3; 0: the caller's stream) side streams, one branch's tail overlaps
        # another's next launch.  The backward runs on the same streams (modules.join_side_streams after it).
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        pool = modules.side_streams("disc", y.device, int(os.environ.get("TTTS_D_STREAMS", "3")))
        main = torch.cuda.current_stream(y.device) if pool else None

        def on(i, fn):
            """fn() on side stream i (after everything queued on the caller's stream so far)."""
            if not pool:
                return fn()
            s = pool[i % len(pool)]
            modules.fork_to(s, main)
            with torch.cuda.stream(s):
                return fn()

        def join():
            for s in pool:
                main.wait_stream(s)

        # Discriminator phase (no gradient flows to y_hat): real and generated clips go through each sub-discriminator as ONE
        # batch of 2B -- the layers are per-sample (no batch statistics), so every output is the same, with half the launches,
        # one weight-norm / operand split per layer instead of two, and fuller tiles on the short DiscriminatorP rows.
        # Generator phase: separate calls, the real branch without an autograd graph (its features are detached constants).
        if not (torch.is_grad_enabled() and y_hat.requires_grad):
            both = torch.cat([y, y_hat], 0)
            outs = [on(i, lambda d=d: d(both)) for i, d in enumerate(self.discriminators)]
            join()
            outs = modules.join_after_backward(outs, y.device, pool)
            for out, fmap in outs:
                y_d_r, y_d_g = out.chunk(2, 0)
                y_d_rs.append(y_d_r)
                y_d_gs.append(y_d_g)
                fmap_rs.append([f.chunk(2, 0)[0] for f in fmap])
                fmap_gs.append([f.chunk(2, 0)[1] for f in fmap])
            return y_d_rs, y_d_gs, fmap_rs, fmap_gs

        def real(d):
            with torch.no_grad():
                return d(y)
        outs = []
        for i, d in enumerate(self.discriminators):
            outs.append((on(i + 1, lambda d=d: real(d)), on(i, lambda d=d: d(y_hat))))
        join()
        outs = modules.join_after_backward(outs, y.device, pool)
        for (y_d_r, fmap_r), (y_d_g, fmap_g) in outs:
            y_d_rs.append(y_d_r)
            y_d_gs.append(y_d_g)
            fmap_rs.append(fmap_r)
            fmap_gs.append(fmap_g)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs


class ResidualCouplingBlock(nn.Module):
    """vq2.py:208-252: n_flows x (mean-only ResidualCouplingLayer, Flip), forward and reverse direction."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4, gin_channels=0):
        super().__init__()
        self.channels, self.hidden_channels, self.kernel_size = channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_layers, self.n_flows, self.gin_channels = dilation_rate, n_layers, n_flows, gin_channels
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(modules.ResidualCouplingLayer(channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                                                            gin_channels=gin_channels, mean_only=True))
            self.flows.append(modules.Flip())

    def forward(self, x, x_mask, g=None, reverse=False):
        if reverse:                                        # vq2.py:249-251
            for flow in reversed(self.flows):
                x = flow(x, x_mask, g=g, reverse=True)
            return x
        for flow in self.flows:
            x, _ = flow(x, x_mask, g=g, reverse=reverse)
        return x


class PosteriorAudioEncoder(nn.Module):
    """vq2.py:667-745: waveform down-conv/ResBlock1 stack (x640) + spectrogram WN stack -> (z, m, logs).
    `noise` (optional, shape of m) replaces the internal randn draw (tests / injected noise)."""

    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0):
        super().__init__()
        self.in_channels, self.out_channels, self.hidden_channels = in_channels, out_channels, hidden_channels
        self.kernel_size, self.dilation_rate, self.n_layers, self.gin_channels = kernel_size, dilation_rate, n_layers, gin_channels
        self.pre = Conv1d(in_channels, hidden_channels, 1)
        self.down_pre = Conv1d(1, 16, 7, 1, padding=3)
        self.resblocks = nn.ModuleList()
        downsample_rates = [10, 8, 2, 2, 2]
        downsample_kernel_sizes = [16, 16, 8, 2, 2]
        ch = [16, 32, 64, 96, 128, 192]
        self.num_kernels = 3
        self.downs = nn.ModuleList()
        for i, (u, k) in enumerate(zip(downsample_rates, downsample_kernel_sizes)):
            self.downs.append(Conv1d(ch[i], ch[i + 1], k, u, padding=(k - 1) // 2).apply_weight_norm("old"))
        for i in range(5):
            for k, d in zip([3, 7, 11], [[1, 3, 5]] * 3):
                self.resblocks.append(modules.ResBlock1(ch[i + 1], k, d))
        self.activation_post = modules.Activation1d(activation=modules.SnakeBeta(ch[-1], alpha_logscale=True))
        self.conv_post = Conv1d(ch[-1], hidden_channels, 7, 1, padding=3)
        self.enc = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.proj = Conv1d(hidden_channels * 2, out_channels * 2, 1)

    def forward(self, x, x_audio, x_mask, g=None, noise=None):
        x_audio = self.down_pre(x_audio)
        for i in range(5):
            x_audio = self.downs[i](x_audio)
            xs = modules.run_branches([lambda j=j: self.resblocks[i * self.num_kernels + j](x_audio) for j in range(self.num_kernels)],
                                      x_audio.device)
            x_audio = modules.add_scale(xs, 1.0 / self.num_kernels)
        x_audio = self.activation_post(x_audio)
        assert x_audio.shape[-1] == x_mask.shape[-1]
        x_audio = self.conv_post(x_audio, omask=x_mask)
        x = self.pre(x, omask=x_mask)
        x = self.enc(x, x_mask, g=g)
        stats = self.proj(torch.cat([x, x_audio], dim=1), omask=x_mask)
        m, logs = torch.split(stats, self.out_channels, dim=1)
        if noise is None:
            noise = torch.randn(m.shape, dtype=m.dtype, device=m.device)
        z = modules._GaussSampleFn.apply(stats, noise, x_mask.reshape(x_mask.shape[0], -1).contiguous())
        return z, m, logs


def sequence_mask(length, max_length=None):
    """commons.sequence_mask (ttts/utils/commons.py:116-120)."""
    if max_length is None:
        max_length = length.max()
    x = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


class _EmbeddingCTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, table):
        ctx.save_for_backward(idx)
        ctx.rows = table.shape[0]
        return ops.embedding_ct_fwd(idx, table)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return None, ops.embedding_ct_bwd(idx, dy, ctx.rows)


class MRTE(nn.Module):
    """vq2.py:17-46: cross-attention of the audio stream (queries) over the text stream (keys / values)."""

    def __init__(self, content_enc_channels=192, hidden_size=512, out_channels=192, kernel_size=5, n_heads=4, ge_layer=2):
        super().__init__()
        self.cross_attention = MultiHeadAttention(hidden_size, hidden_size, n_heads)
        self.c_pre = Conv1d(content_enc_channels, hidden_size, 1)
        self.text_pre = Conv1d(content_enc_channels, hidden_size, 1)
        self.c_post = Conv1d(hidden_size, out_channels, 1)

    def forward(self, ssl_enc, ssl_mask, text, text_mask, ge, test=None):
        ssl_enc = self.c_pre(modules.mul_mask(ssl_enc, ssl_mask))
        text_enc = self.text_pre(modules.mul_mask(text, text_mask))
        # (cross_attention(..) + ssl_enc + ge) * ssl_mask: residual, broadcast style vector and mask are epilogues of the
        # attention's output projection
        x = self.cross_attention(modules.mul_mask(ssl_enc, ssl_mask), modules.mul_mask(text_enc, text_mask),
                                 q_mask=ssl_mask, k_mask=text_mask, resid=ssl_enc,
                                 bbias=ge.squeeze(-1) if ge is not None else None, omask=ssl_mask)
        return self.c_post(x)


class TextEncoder(nn.Module):
    """vq2.py:90-164."""

    def __init__(self, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                 latent_channels=192):
        super().__init__()
        self.out_channels, self.hidden_channels, self.filter_channels = out_channels, hidden_channels, filter_channels
        self.n_heads, self.n_layers, self.kernel_size, self.p_dropout = n_heads, n_layers, kernel_size, p_dropout
        self.latent_channels = latent_channels
        self.encoder_ssl = attentions.Encoder(hidden_channels, filter_channels, n_heads, n_layers // 2, kernel_size, p_dropout)
        self.encoder_text = attentions.Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.text_embedding = nn.Embedding(256, hidden_channels)
        self.mrte = MRTE()
        self.encoder2 = attentions.Encoder(hidden_channels, filter_channels, n_heads, n_layers // 2, kernel_size, p_dropout)
        self.proj = Conv1d(hidden_channels, out_channels * 2, 1)

    def forward(self, y, y_lengths, text, text_lengths, ge, test=None):
        y_mask = torch.unsqueeze(sequence_mask(y_lengths, y.size(2)), 1).to(y.dtype)
        y = self.encoder_ssl(y, y_mask)                 # Encoder masks its input itself (x * x_mask)
        text_mask = torch.unsqueeze(sequence_mask(text_lengths, text.size(1)), 1).to(y.dtype)
        if test == 1:
            text = torch.zeros_like(text)
        text = _EmbeddingCTFn.apply(text, self.text_embedding.weight)
        text = self.encoder_text(text, text_mask)
        y = self.mrte(y, y_mask, text, text_mask, ge)
        y = self.encoder2(y, y_mask)
        stats = self.proj(y, omask=y_mask)
        m, logs = torch.split(stats, self.out_channels, dim=1)
        return y, m, logs


def slice_segments(x, ids_str, segment_size=4):
    """commons.slice_segments (ttts/utils/commons.py:48-54) as one gather instead of a Python loop over the batch."""
    idx = ids_str.view(-1, 1) + torch.arange(segment_size, device=x.device).view(1, -1)            # (B, seg)
    return torch.gather(x, 2, idx.unsqueeze(1).expand(-1, x.size(1), -1))


def rand_slice_segments(x, x_lengths=None, segment_size=4, ids_str=None):
    """commons.rand_slice_segments (:57-66); `ids_str` injects the window starts (tests)."""
    b, d, t = x.size()
    if x_lengths is None:
        x_lengths = t
    if ids_str is None:
        ids_str_max = x_lengths - segment_size + 1
        ids_str = (torch.rand([b], device=x.device) * ids_str_max).to(dtype=torch.long)
    return slice_segments(x, ids_str, segment_size), ids_str


def _prior_sample(m_p, logs_p, noise, noise_scale):
    """z_p = m_p + randn * exp(logs_p) * noise_scale (vq2.py:886,903), the reparameterised-sample kernel of the training path."""
    from .. import ops
    eps = torch.randn_like(m_p) if noise is None else noise
    return ops.gauss_sample_fwd(torch.cat([m_p, logs_p], 1), eps * noise_scale, None)


class SynthesizerTrn(nn.Module):
    """Synthesizer for training (vq2.py:750-871): same constructor, `forward(wav, wav_aug, wav_lengths, y, y_aug,
    y_lengths, text, text_lengths)` and 6-tuple return.  Extra keyword-only hooks `noise_p`, `noise_q`, `ids_slice`
    inject the three random draws (posterior samples, segment starts) for parity tests."""

    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads, n_layers,
                 kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, prosody_size=20, n_speakers=0, gin_channels=0,
                 semantic_frame_rate=None, freeze_quantizer=None, **kwargs):
        super().__init__()
        self.spec_channels, self.inter_channels, self.hidden_channels = spec_channels, inter_channels, hidden_channels
        self.filter_channels, self.n_heads, self.n_layers, self.kernel_size = filter_channels, n_heads, n_layers, kernel_size
        self.p_dropout, self.resblock = p_dropout, resblock
        self.segment_size, self.n_speakers, self.gin_channels, self.mel_size = segment_size, n_speakers, gin_channels, prosody_size
        self.dec = Generator(inter_channels, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                             upsample_initial_channel, upsample_kernel_sizes, gin_channels=gin_channels)
        self.enc_p = PosteriorAudioEncoder(spec_channels, inter_channels, hidden_channels, 5, 1, 16, gin_channels=gin_channels)
        self.enc_p_2 = TextEncoder(inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.enc_q = PosteriorAudioEncoder(spec_channels, inter_channels, hidden_channels, 5, 1, 16, gin_channels=gin_channels)
        self.flow = ResidualCouplingBlock(inter_channels, hidden_channels, 5, 1, 4, gin_channels=gin_channels)
        self.ref_enc = MelStyleEncoder(spec_channels, style_vector_dim=gin_channels)
        self.quantizer = ResidualVectorQuantizer(dimension=inter_channels, n_q=1, bins=1024)
        self.proj = Conv1d(inter_channels, inter_channels, 2, stride=2)
        if freeze_quantizer:
            self.enc_p.requires_grad_(False)

    def forward(self, wav, wav_aug, wav_lengths, y, y_aug, y_lengths, text, text_lengths, *, noise_p=None, noise_q=None,
                ids_slice=None):
        y_mask = torch.unsqueeze(sequence_mask(y_lengths, y.size(2)), 1).to(y.dtype)
        ge = self.ref_enc(modules.mul_mask(y, y_mask), y_mask)
        # the prior path (augmented clip -> codes -> text-conditioned prior) and the posterior / flow / decoder path only share
        # the style vector: two concurrent branches.  The randn draws happen here, in the reference's order (enc_p first).
        if noise_p is None:
            noise_p = torch.randn(y.size(0), self.inter_channels, y.size(2), dtype=y.dtype, device=y.device)
        if noise_q is None:
            noise_q = torch.randn(y.size(0), self.inter_channels, y.size(2), dtype=y.dtype, device=y.device)

        def prior():
            x, _, _ = self.enc_p(y_aug, wav_aug.unsqueeze(1), y_mask, g=ge, noise=noise_p)
            x = self.proj(x)
            quantized, codes, commit_loss, quantized_list = self.quantizer(x, layers=[0])
            quantized = modules.upsample_nearest2(quantized)
            x, m_p, logs_p = self.enc_p_2(quantized, y_lengths, text, text_lengths, ge)
            return quantized, commit_loss, m_p, logs_p

        def posterior():
            z, m_q, logs_q = self.enc_q(y, wav.unsqueeze(1), y_mask, g=ge, noise=noise_q)
            z_p = self.flow(z, y_mask, g=ge)
            z_slice, ids = rand_slice_segments(z, y_lengths, self.segment_size, ids_slice)
            return z, m_q, logs_q, z_p, ids, self.dec(z_slice, g=ge)

        (quantized, commit_loss, m_p, logs_p), (z, m_q, logs_q, z_p, ids_slice, o) = modules.run_branches(
            [prior, posterior], y.device, pool="synth", inline=_SYNTH_INLINE)   # (the decoder's own fan-out forks from this stream)
        return o, commit_loss, ids_slice, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q), quantized

    @torch.no_grad()
    def infer(self, wav, wav_lengths, y, y_lengths, text, text_lengths, noise_scale=0.5, *, noise_p=None, noise=None):
        """vq2.py:873-889: re-synthesis of a clip from its own codes -- posterior -> codebook -> text encoder prior -> sample ->
        REVERSE flow -> decoder over the whole clip.  `noise_p` / `noise` inject the two randn draws (parity tests)."""
        y_mask = torch.unsqueeze(sequence_mask(y_lengths, y.size(2)), 1).to(y.dtype)
        ge = self.ref_enc(modules.mul_mask(y, y_mask), y_mask)
        x, _, _ = self.enc_p(y, wav.unsqueeze(1), y_mask, g=ge, noise=noise_p)
        x = self.proj(x)
        quantized, codes, commit_loss, quantized_list = self.quantizer(x, layers=[0])
        quantized = modules.upsample_nearest2(quantized)
        x, m_p, logs_p = self.enc_p_2(quantized, y_lengths, text, text_lengths, ge)
        z_p = _prior_sample(m_p, logs_p, noise, noise_scale)
        z = self.flow(z_p, y_mask, g=ge, reverse=True)
        return self.dec(z, g=ge)

    @torch.no_grad()
    def decode(self, codes, text, refer, noise_scale=0.5, *, noise=None):
        """vq2.py:891-910 as its body intends: codes (n_q, 1, T) + text (1, Tt) + a reference spectrogram (1, spec_channels, Tr)
        for the style vector -> waveform (1, 1, 2 T hop).  The reference body is not runnable as written (undefined
        `text_legnths` / `y_mask`, and `y_lengths` taken before the x2 upsampling: SURVEY.md App. B); here the lengths are those
        of the upsampled code sequence and the mask is all ones, which is what a single un-padded item means."""
        refer_lengths = torch.tensor([refer.size(2)], dtype=torch.long, device=refer.device)
        text_lengths = torch.tensor([text.size(1)], dtype=torch.long, device=text.device)
        refer_mask = torch.unsqueeze(sequence_mask(refer_lengths, refer.size(2)), 1).to(refer.dtype)
        ge = self.ref_enc(modules.mul_mask(refer, refer_mask), refer_mask)
        quantized = modules.upsample_nearest2(self.quantizer.decode(codes))
        y_lengths = torch.tensor([quantized.size(2)], dtype=torch.long, device=codes.device)
        y_mask = torch.unsqueeze(sequence_mask(y_lengths, quantized.size(2)), 1).to(refer.dtype)
        x, m_p, logs_p = self.enc_p_2(quantized, y_lengths, text, text_lengths, ge)
        z_p = _prior_sample(m_p, logs_p, noise, noise_scale)
        z = self.flow(z_p, y_mask, g=ge, reverse=True)
        return self.dec(modules.mul_mask(z, y_mask), g=ge)

    @torch.no_grad()
    def extract_latent(self, wav, y, y_lengths=None):
        """Code extraction (vq2.py:912-920): ref_enc -> enc_p -> proj -> quantizer; returns codes (B, n_q, T // 2) int64.
        The reference body uses an undefined `y_lengths` and multiplies the stride-2 projection by the full-rate mask
        (SURVEY.md App. B), so this follows the training forward instead: lengths default to the full clip and there is
        no mask around `proj`.  The reference also quantises a SAMPLED posterior (enc_p draws randn noise even here);
        extraction uses the posterior mean (zero noise) so that the stored codes are reproducible."""
        if y_lengths is None:
            y_lengths = torch.full((y.size(0),), y.size(2), dtype=torch.long, device=y.device)
        y_mask = torch.unsqueeze(sequence_mask(y_lengths, y.size(2)), 1).to(y.dtype)
        ge = self.ref_enc(modules.mul_mask(y, y_mask), y_mask)
        x, _, _ = self.enc_p(y, wav.unsqueeze(1), y_mask, g=ge, noise=torch.zeros(y.size(0), self.inter_channels, y.size(2),
                                                                                 dtype=y.dtype, device=y.device))
        x = self.proj(x)
        quantized, codes, commit_loss, quantized_list = self.quantizer(x)
        return codes.transpose(0, 1)
