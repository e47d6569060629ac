"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the conv stacks of the VQ-VAE-GAN path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Restates (reference = /root/reference, adelacvg/ttts), as functions of a state dict with the reference's keys:
* `ResBlock1.forward`        ttts/vqvae/modules.py:224-318  (6 weight-normed dilated convs, leaky-relu 0.1, residual)
* `Generator.forward`        ttts/vqvae/vq2.py:341-415      (HiFi-GAN decoder: conv_pre, cond, ConvTranspose1d ups,
                                                              resblock sums / num_kernels, conv_post, tanh)
* weight norm in both styles ttts/vqvae/vq2.py:10,364 (`weight_g/weight_v`), modules.py:8 (`parametrizations.weight.
  original0/original1`); `get_padding` ttts/utils/commons.py:12-13.

Parity pin: `tests/golden/vqvae_generator.npz` generated from the imported reference by `tools/make_goldens.py`.
"""
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # ttts/vqvae/modules.py:16


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


def _wn(v, g):
    return g * v / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))


def _weight(sd, pfx):
    """Weight of a (possibly weight-normed) conv stored under either naming style."""
    if pfx + "weight" in sd:
        return sd[pfx + "weight"]
    if pfx + "weight_v" in sd:
        return _wn(sd[pfx + "weight_v"], sd[pfx + "weight_g"])
    return _wn(sd[pfx + "parametrizations.weight.original1"], sd[pfx + "parametrizations.weight.original0"])


def resblock1(x, sd, pfx, kernel_size, dilation):
    for n, d in enumerate(dilation):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _weight(sd, f"{pfx}convs1.{n}."), sd[f"{pfx}convs1.{n}.bias"], dilation=d,
                      padding=get_padding(kernel_size, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, _weight(sd, f"{pfx}convs2.{n}."), sd[f"{pfx}convs2.{n}.bias"], padding=get_padding(kernel_size, 1))
        x = xt + x
    return x


def generator_forward(sd, cfg, x, g=None, pfx=""):
    """cfg: resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates, upsample_kernel_sizes."""
    nk = len(cfg["resblock_kernel_sizes"])
    x = F.conv1d(x, sd[pfx + "conv_pre.weight"], sd[pfx + "conv_pre.bias"], padding=3)
    if g is not None:
        x = x + F.conv1d(g, sd[pfx + "cond.weight"], sd[pfx + "cond.bias"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, _weight(sd, f"{pfx}ups.{i}."), sd[f"{pfx}ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (ks, ds) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            r = resblock1(x, sd, f"{pfx}resblocks.{i * nk + j}.", ks, ds)
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)            # default slope 0.01 (vq2.py:404)
    x = F.conv1d(x, sd[pfx + "conv_post.weight"], None, padding=3)
    return torch.tanh(x)


# ---- discriminators (ttts/vqvae/vq2.py:418-551) and losses (ttts/vqvae/losses.py:7-61) --------------------------------
def det_fill(name, shape, gain=1.0):
    """Deterministic, construction-order-independent parameter fill shared by tools/make_goldens.py, the oracle tests and
    the GPU parity tests (so multi-million-parameter state dicts need not be stored).  `gain` < 1 makes every layer
    contractive (used for the full-model fixtures, where unit-gain residual stacks would blow up)."""
    import zlib
    import numpy as np
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    a = rng.standard_normal(size=tuple(shape), dtype=np.float32)
    if name.endswith("weight_g") or name.endswith("original0"):
        a = np.float32(gain) * (np.float32(1.0) + np.float32(0.1) * np.abs(a))
    elif len(shape) > 1:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        a = a * np.float32(gain * 1.2 / fan_in ** 0.5)
    elif name.endswith("gamma"):
        a = np.float32(1.0) + a * np.float32(0.05) if gain != 1.0 else a * np.float32(0.05)
    else:
        a = a * np.float32(0.05)
    return torch.from_numpy(a)


DS_SPEC = [(15, 1, 1, 7), (41, 4, 4, 20), (41, 4, 16, 20), (41, 4, 64, 20), (41, 4, 256, 20), (5, 1, 1, 2)]  # k, stride, groups, pad


def discriminator_s(x, sd, pfx):
    fmap = []
    for i, (k, s, g, pd) in enumerate(DS_SPEC):
        x = F.leaky_relu(F.conv1d(x, _weight(sd, f"{pfx}convs.{i}."), sd[f"{pfx}convs.{i}.bias"], stride=s, padding=pd, groups=g),
                         LRELU_SLOPE)
        fmap.append(x)
    x = F.conv1d(x, _weight(sd, pfx + "conv_post."), sd[pfx + "conv_post.bias"], padding=1)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def discriminator_p(x, sd, pfx, period, kernel_size=5, stride=3):
    fmap = []
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), "reflect")
        t = t + n_pad
    x = x.view(b, c, t // period, period)
    pad = (get_padding(kernel_size, 1), 0)
    for i in range(5):
        x = F.leaky_relu(F.conv2d(x, _weight(sd, f"{pfx}convs.{i}."), sd[f"{pfx}convs.{i}.bias"],
                                  stride=(stride if i < 4 else 1, 1), padding=pad), LRELU_SLOPE)
        fmap.append(x)
    x = F.conv2d(x, _weight(sd, pfx + "conv_post."), sd[pfx + "conv_post.bias"], padding=(1, 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


PERIODS = [2, 3, 5, 7, 11]


def mpd_forward(sd, y, y_hat):
    outs = ([], [], [], [])
    for i in range(6):
        pfx = f"discriminators.{i}."
        f = (lambda t: discriminator_s(t, sd, pfx)) if i == 0 else (lambda t: discriminator_p(t, sd, pfx, PERIODS[i - 1]))
        r, fr = f(y)
        g, fg = f(y_hat)
        outs[0].append(r); outs[1].append(g); outs[2].append(fr); outs[3].append(fg)
    return outs


def feature_loss(fmap_r, fmap_g):
    loss = 0
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean(torch.abs(rl.detach() - gl))
    return loss * 2


def discriminator_loss(dr_list, dg_list):
    loss = 0
    for dr, dg in zip(dr_list, dg_list):
        loss = loss + torch.mean((1 - dr) ** 2) + torch.mean(dg ** 2)
    return loss


def generator_loss(dg_list):
    loss = 0
    for dg in dg_list:
        loss = loss + torch.mean((1 - dg) ** 2)
    return loss


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    kl = logs_p - logs_q - 0.5
    kl = kl + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2.0 * logs_p)
    return torch.sum(kl * z_mask) / torch.sum(z_mask)


# ---- WN / coupling flow / anti-aliased SnakeBeta / PosteriorAudioEncoder ---------------------------------------------------
def wn_forward(x, x_mask, sd, pfx, hidden, kernel_size, dilation_rate, n_layers, g=None):
    """modules.WN.forward (ttts/vqvae/modules.py:187-213), p_dropout = 0."""
    output = torch.zeros_like(x)
    if g is not None:
        g = F.conv1d(g, _weight(sd, pfx + "cond_layer."), sd[pfx + "cond_layer.bias"])
    for i in range(n_layers):
        dil = dilation_rate ** i
        pad = int((kernel_size * dil - dil) / 2)
        x_in = F.conv1d(x, _weight(sd, f"{pfx}in_layers.{i}."), sd[f"{pfx}in_layers.{i}.bias"], dilation=dil, padding=pad)
        if g is not None:
            x_in = x_in + g[:, i * 2 * hidden:(i + 1) * 2 * hidden, :]
        acts = torch.tanh(x_in[:, :hidden]) * torch.sigmoid(x_in[:, hidden:])
        rs = F.conv1d(acts, _weight(sd, f"{pfx}res_skip_layers.{i}."), sd[f"{pfx}res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * x_mask


def coupling_block_forward(x, x_mask, sd, pfx, channels, hidden, kernel_size, dilation_rate, n_layers, n_flows, g=None):
    """ResidualCouplingBlock.forward (vq2.py:245-252) with mean-only layers (modules.py:440-459) and Flip."""
    half = channels // 2
    for f in range(n_flows):
        p = f"{pfx}flows.{2 * f}."
        x0, x1 = torch.split(x, [half, half], 1)
        h = F.conv1d(x0, sd[p + "pre.weight"], sd[p + "pre.bias"]) * x_mask
        h = wn_forward(h, x_mask, sd, p + "enc.", hidden, kernel_size, dilation_rate, n_layers, g)
        m = F.conv1d(h, sd[p + "post.weight"], sd[p + "post.bias"]) * x_mask
        x = torch.cat([x0, m + x1 * x_mask], 1)
        x = torch.flip(x, [1])
    return x


def coupling_block_reverse(x, x_mask, sd, pfx, channels, hidden, kernel_size, dilation_rate, n_layers, n_flows, g=None):
    """ResidualCouplingBlock.forward(reverse=True) (vq2.py:245-252: the flows in reversed order; Flip first, then the mean-only
    layer's inverse x1 = (x1 - m) * mask, modules.py:456-459)."""
    half = channels // 2
    for f in reversed(range(n_flows)):
        p = f"{pfx}flows.{2 * f}."
        x = torch.flip(x, [1])
        x0, x1 = torch.split(x, [half, half], 1)
        h = F.conv1d(x0, sd[p + "pre.weight"], sd[p + "pre.bias"]) * x_mask
        h = wn_forward(h, x_mask, sd, p + "enc.", hidden, kernel_size, dilation_rate, n_layers, g)
        m = F.conv1d(h, sd[p + "post.weight"], sd[p + "post.bias"]) * x_mask
        x = torch.cat([x0, (x1 - m) * x_mask], 1)
    return x


def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """alias_free_torch/filter.py:28-56."""
    import math
    half = kernel_size // 2
    A = 2.285 * (half - 1) * math.pi * 4 * half_width + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half, half) + 0.5) if kernel_size % 2 == 0 else torch.arange(kernel_size) - half
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (f / f.sum()).view(1, 1, kernel_size)


def snake_aa(x, alpha, beta, fup=None, fdn=None):
    """Activation1d(SnakeBeta(alpha_logscale=True)) with ratio 2, 12 taps (alias_free_torch/act.py, resample.py;
    activations.py:101-119)."""
    C = x.shape[1]
    fup = kaiser_sinc_filter1d(0.25, 0.3, 12) if fup is None else fup
    fdn = kaiser_sinc_filter1d(0.25, 0.3, 12) if fdn is None else fdn
    u = F.pad(x, (5, 5), mode="replicate")
    u = 2 * F.conv_transpose1d(u, fup.expand(C, -1, -1), stride=2, groups=C)[..., 15:-15]
    a, b = torch.exp(alpha)[None, :, None], torch.exp(beta)[None, :, None]
    v = u + (1.0 / (b + 1e-9)) * torch.sin(u * a) ** 2
    v = F.pad(v, (5, 6), mode="replicate")
    return F.conv1d(v, fdn.expand(C, -1, -1), stride=2, groups=C)


def posterior_audio_encoder_forward(sd, pfx, x, x_audio, x_mask, g, noise, out_channels=192, hidden=192):
    """PosteriorAudioEncoder.forward (vq2.py:714-745) with the randn draw injected."""
    ks, ss = [16, 16, 8, 2, 2], [10, 8, 2, 2, 2]
    xa = F.conv1d(x_audio, sd[pfx + "down_pre.weight"], sd[pfx + "down_pre.bias"], padding=3)
    for i in range(5):
        xa = F.conv1d(xa, _weight(sd, f"{pfx}downs.{i}."), sd[f"{pfx}downs.{i}.bias"], stride=ss[i], padding=(ks[i] - 1) // 2)
        xs = None
        for j, k in enumerate([3, 7, 11]):
            r = resblock1(xa, sd, f"{pfx}resblocks.{i * 3 + j}.", k, (1, 3, 5))
            xs = r if xs is None else xs + r
        xa = xs / 3
    xa = snake_aa(xa, sd[pfx + "activation_post.act.alpha"], sd[pfx + "activation_post.act.beta"])
    xa = F.conv1d(xa, sd[pfx + "conv_post.weight"], sd[pfx + "conv_post.bias"], padding=3)
    h = F.conv1d(x, sd[pfx + "pre.weight"], sd[pfx + "pre.bias"]) * x_mask
    h = wn_forward(h, x_mask, sd, pfx + "enc.", hidden, 5, 1, 16, g)
    xa = xa * x_mask
    stats = F.conv1d(torch.cat([h, xa], 1), sd[pfx + "proj.weight"], sd[pfx + "proj.bias"]) * x_mask
    m, logs = torch.split(stats, out_channels, dim=1)
    z = (m + noise * torch.exp(logs)) * x_mask
    return z, m, logs


# ---- attention stacks: attentions.Encoder / MultiHeadAttention / FFN, MRTE, TextEncoder, MelStyleEncoder ---------------------
def _ln_ch(x, sd, pfx, eps=1e-5):
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), sd[pfx + "gamma"], sd[pfx + "beta"], eps).transpose(1, -1)


def mha_forward(x, c, attn_mask, sd, pfx, n_heads, window):
    """attentions.MultiHeadAttention.forward / attention (attentions.py:231-289); dropout off.  The relative-position
    terms are written as explicit diagonals instead of the reference's pad-and-reshape skewing (:320-356)."""
    import math
    q = F.conv1d(x, sd[pfx + "conv_q.weight"], sd[pfx + "conv_q.bias"])
    k = F.conv1d(c, sd[pfx + "conv_k.weight"], sd[pfx + "conv_k.bias"])
    v = F.conv1d(c, sd[pfx + "conv_v.weight"], sd[pfx + "conv_v.bias"])
    b, d, t_s = k.shape
    t_t = q.shape[2]
    dk = d // n_heads
    q = q.view(b, n_heads, dk, t_t).transpose(2, 3)
    k = k.view(b, n_heads, dk, t_s).transpose(2, 3)
    v = v.view(b, n_heads, dk, t_s).transpose(2, 3)
    qs = q / math.sqrt(dk)
    scores = torch.matmul(qs, k.transpose(-2, -1))
    if window:
        ek, ev = sd[pfx + "emb_rel_k"], sd[pfx + "emb_rel_v"]          # (1, 2w+1, dk)
        idx = torch.arange(t_t)
        rel = idx[None, :] - idx[:, None]                               # j - i
        inside = (rel.abs() <= window)
        r = (rel + window).clamp(0, 2 * window)
        logits = torch.einsum("bhid,hrd->bhir", qs, ek.expand(n_heads, -1, -1) if ek.shape[0] == 1 else ek)
        scores = scores + torch.gather(logits, 3, r[None, None].expand(b, n_heads, -1, -1)) * inside
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask == 0, -1e4)
    p = F.softmax(scores, dim=-1)
    out = torch.matmul(p, v)
    if window:
        evh = ev.expand(n_heads, -1, -1) if ev.shape[0] == 1 else ev
        pw = torch.zeros(b, n_heads, t_t, 2 * window + 1)
        for rr in range(2 * window + 1):
            diag = torch.diagonal(p, offset=rr - window, dim1=2, dim2=3)        # p[i, i + rr - w]
            lo = max(0, window - rr)
            pw[:, :, lo:lo + diag.shape[-1], rr] = diag
        out = out + torch.einsum("bhir,hrd->bhid", pw, evh)
    out = out.transpose(2, 3).contiguous().view(b, d, t_t)
    return F.conv1d(out, sd[pfx + "conv_o.weight"], sd[pfx + "conv_o.bias"])


def ffn_forward(x, x_mask, sd, pfx, kernel_size):
    pad = ((kernel_size - 1) // 2, kernel_size // 2)
    x = F.conv1d(F.pad(x * x_mask, pad), sd[pfx + "conv_1.weight"], sd[pfx + "conv_1.bias"])
    x = torch.relu(x)
    x = F.conv1d(F.pad(x * x_mask, pad), sd[pfx + "conv_2.weight"], sd[pfx + "conv_2.bias"])
    return x * x_mask


def encoder_forward(x, x_mask, sd, pfx, n_heads, n_layers, kernel_size, window=4):
    """attentions.Encoder.forward (attentions.py:66-88), dropout off."""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    for i in range(n_layers):
        y = mha_forward(x, x, attn_mask, sd, f"{pfx}attn_layers.{i}.", n_heads, window)
        x = _ln_ch(x + y, sd, f"{pfx}norm_layers_1.{i}.")
        y = ffn_forward(x, x_mask, sd, f"{pfx}ffn_layers.{i}.", kernel_size)
        x = _ln_ch(x + y, sd, f"{pfx}norm_layers_2.{i}.")
    return x * x_mask


def seq_mask(lengths, max_len):
    return (torch.arange(max_len)[None, :] < lengths[:, None])


def text_encoder_forward(sd, pfx, y, y_lengths, text, text_lengths, ge, n_heads=2, n_layers=6, kernel_size=3, out_channels=192):
    """TextEncoder.forward (vq2.py:144-164) + MRTE.forward (vq2.py:33-46), dropout off."""
    y_mask = seq_mask(y_lengths, y.size(2)).unsqueeze(1).to(y.dtype)
    y = encoder_forward(y * y_mask, y_mask, sd, pfx + "encoder_ssl.", n_heads, n_layers // 2, kernel_size)
    text_mask = seq_mask(text_lengths, text.size(1)).unsqueeze(1).to(y.dtype)
    t = F.embedding(text, sd[pfx + "text_embedding.weight"]).transpose(1, 2)
    t = encoder_forward(t * text_mask, text_mask, sd, pfx + "encoder_text.", n_heads, n_layers, kernel_size)
    m = pfx + "mrte."
    attn_mask = text_mask.unsqueeze(2) * y_mask.unsqueeze(-1)
    ssl_enc = F.conv1d(y * y_mask, sd[m + "c_pre.weight"], sd[m + "c_pre.bias"])
    text_enc = F.conv1d(t * text_mask, sd[m + "text_pre.weight"], sd[m + "text_pre.bias"])
    x = mha_forward(ssl_enc * y_mask, text_enc * text_mask, attn_mask, sd, m + "cross_attention.", 4, 0) + ssl_enc + ge
    y = F.conv1d(x * y_mask, sd[m + "c_post.weight"], sd[m + "c_post.bias"])
    y = encoder_forward(y * y_mask, y_mask, sd, pfx + "encoder2.", n_heads, n_layers // 2, kernel_size)
    stats = F.conv1d(y, sd[pfx + "proj.weight"], sd[pfx + "proj.bias"]) * y_mask
    m_, logs = torch.split(stats, out_channels, dim=1)
    return y, m_, logs


def mel_style_encoder_forward(sd, pfx, x, mask, hidden=128, n_head=2):
    """MelStyleEncoder.forward (modules.py:736-764), eval mode; x (B, n_mel, T), mask (B, 1, T) 1 = valid."""
    x = x.transpose(1, 2)
    pad = (mask.int() == 0).squeeze(1)
    lin = lambda t, k: F.linear(t, sd[pfx + k + ".weight"], sd[pfx + k + ".bias"])
    x = F.mish(lin(x, "spectral.0.fc"))
    x = F.mish(lin(x, "spectral.3.fc"))
    x = x.transpose(1, 2)
    for i in range(2):
        h = F.conv1d(x, sd[f"{pfx}temporal.{i}.conv1.conv.weight"], sd[f"{pfx}temporal.{i}.conv1.conv.bias"], padding=2)
        x = x + h[:, :hidden] * torch.sigmoid(h[:, hidden:])
    x = x.transpose(1, 2)
    x = x.masked_fill(pad.unsqueeze(-1), 0)
    b, t, _ = x.shape
    dk = hidden // n_head
    split = lambda z: z.view(b, t, n_head, dk).permute(2, 0, 1, 3).reshape(-1, t, dk)
    q, k, v = split(lin(x, "slf_attn.w_qs")), split(lin(x, "slf_attn.w_ks")), split(lin(x, "slf_attn.w_vs"))
    attn = torch.bmm(q, k.transpose(1, 2)) / (hidden ** 0.5)
    attn = attn.masked_fill(pad.unsqueeze(1).expand(-1, t, -1).repeat(n_head, 1, 1), -float("inf"))
    out = torch.bmm(torch.softmax(attn, dim=2), v)
    out = out.view(n_head, b, t, dk).permute(1, 2, 0, 3).reshape(b, t, -1)
    x = lin(out, "slf_attn.fc") + x
    x = lin(x, "fc.fc")
    x = x.masked_fill(pad.unsqueeze(-1), 0).sum(dim=1) / (~pad).sum(dim=1).unsqueeze(1)
    return x.unsqueeze(-1)


# ---- SynthesizerTrn.forward (vq2.py:842-871): the composition of everything above ------------------------------------------
def synthesizer_forward(sd, cfg, buffers, wav, wav_aug, wav_lengths, y, y_aug, y_lengths, text, text_lengths, noise_p, noise_q,
                        ids_slice, segment_size=32, training=True):
    """sd: reference-keyed generator state dict; cfg: the `vqvae` block of vqvae/config.json; buffers: the codebook buffers
    dict(embed, embed_avg, cluster_size) (updated in place when training); the three random draws are injected.
    Returns the reference's 6-tuple (o, commit_loss, ids_slice, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q), quantized)."""
    from oracle import vq_ref
    y_mask = seq_mask(y_lengths, y.size(2)).unsqueeze(1).to(y.dtype)
    ge = mel_style_encoder_forward(sd, "ref_enc.", y * y_mask, y_mask)
    x, _, _ = posterior_audio_encoder_forward(sd, "enc_p.", y_aug, wav_aug.unsqueeze(1), y_mask, ge, noise_p)
    x = F.conv1d(x, sd["proj.weight"], sd["proj.bias"], stride=2)
    quantized, codes, commit_loss, _ = vq_ref.rvq_forward(x, buffers, training)
    quantized = F.interpolate(quantized, size=int(quantized.shape[-1] * 2), mode="nearest")
    _, m_p, logs_p = text_encoder_forward(sd, "enc_p_2.", quantized, y_lengths, text, text_lengths, ge, cfg["n_heads"],
                                          cfg["n_layers"], cfg["kernel_size"], cfg["inter_channels"])
    z, m_q, logs_q = posterior_audio_encoder_forward(sd, "enc_q.", y, wav.unsqueeze(1), y_mask, ge, noise_q)
    z_p = coupling_block_forward(z, y_mask, sd, "flow.", cfg["inter_channels"], cfg["hidden_channels"], 5, 1, 4, 4, ge)
    idx = ids_slice.view(-1, 1) + torch.arange(segment_size).view(1, -1)
    z_slice = torch.gather(z, 2, idx.unsqueeze(1).expand(-1, z.size(1), -1))
    o = generator_forward(sd, cfg, z_slice, ge, pfx="dec.")
    return o, commit_loss, ids_slice, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q), quantized


def synthesizer_infer(sd, cfg, buffers, wav, wav_lengths, y, y_lengths, text, text_lengths, noise_p, noise, noise_scale=0.5):
    """SynthesizerTrn.infer (vq2.py:873-889), eval mode: quantise the posterior of the clip itself, sample the prior with
    `noise`, run the flow backwards, decode the WHOLE clip.  The two random draws are injected."""
    from oracle import vq_ref
    y_mask = seq_mask(y_lengths, y.size(2)).unsqueeze(1).to(y.dtype)
    ge = mel_style_encoder_forward(sd, "ref_enc.", y * y_mask, y_mask)
    x, _, _ = posterior_audio_encoder_forward(sd, "enc_p.", y, wav.unsqueeze(1), y_mask, ge, noise_p)
    x = F.conv1d(x, sd["proj.weight"], sd["proj.bias"], stride=2)
    quantized, _, _, _ = vq_ref.rvq_forward(x, buffers, False)
    quantized = F.interpolate(quantized, size=int(quantized.shape[-1] * 2), mode="nearest")
    _, m_p, logs_p = text_encoder_forward(sd, "enc_p_2.", quantized, y_lengths, text, text_lengths, ge, cfg["n_heads"],
                                          cfg["n_layers"], cfg["kernel_size"], cfg["inter_channels"])
    z_p = m_p + noise * torch.exp(logs_p) * noise_scale
    z = coupling_block_reverse(z_p, y_mask, sd, "flow.", cfg["inter_channels"], cfg["hidden_channels"], 5, 1, 4, 4, ge)
    return generator_forward(sd, cfg, z, ge, pfx="dec.")


def synthesizer_decode(sd, cfg, buffers, codes, text, refer, noise, noise_scale=0.5):
    """SynthesizerTrn.decode (vq2.py:891-910) as its body intends (it is not runnable as written: undefined `text_legnths` and
    `y_mask`, and y_lengths taken before the x2 upsampling -- SURVEY App. B): codes (n_q, 1, T) -> codebook rows -> x2 nearest ->
    enc_p_2 -> sample -> reverse flow -> dec, with the style vector of `refer` (1, spec_channels, Tr)."""
    refer_mask = torch.ones(1, 1, refer.size(2), dtype=refer.dtype)
    ge = mel_style_encoder_forward(sd, "ref_enc.", refer * refer_mask, refer_mask)
    quantized = F.embedding(codes[0], buffers["embed"]).transpose(1, 2)      # n_q = 1: the layer's codebook rows, (1, D, T)
    quantized = F.interpolate(quantized, size=int(quantized.shape[-1] * 2), mode="nearest")
    y_lengths = torch.tensor([quantized.size(2)])
    y_mask = torch.ones(1, 1, quantized.size(2), dtype=refer.dtype)
    _, m_p, logs_p = text_encoder_forward(sd, "enc_p_2.", quantized, y_lengths, text, torch.tensor([text.size(1)]), ge, cfg["n_heads"],
                                          cfg["n_layers"], cfg["kernel_size"], cfg["inter_channels"])
    z_p = m_p + noise * torch.exp(logs_p) * noise_scale
    z = coupling_block_reverse(z_p, y_mask, sd, "flow.", cfg["inter_channels"], cfg["hidden_channels"], 5, 1, 4, 4, ge)
    return generator_forward(sd, cfg, z * y_mask, ge, pfx="dec.")


# ---- the two-phase GAN step body (ttts/vqvae/train.py:313-406) as loss functions over state dicts --------------------------
def gan_step_losses(sd_g, sd_d, cfg, hps, buffers, wav, wav_lengths, text, text_lengths, noise_p, noise_q, ids_slice,
                    d_update=None):
    """One training step up to the generator loss: spectrogram, SynthesizerTrn.forward, mel of the slice and of y_hat,
    MultiPeriodDiscriminator on (y, y_hat.detach()) -> loss_disc -> `d_update(loss_disc)` (the caller's discriminator phase:
    backward, grad norm, AdamW step on the leaves of `sd_d` IN PLACE, train.py:365-369) -> MultiPeriodDiscriminator again on
    (y, y_hat) with the UPDATED discriminator -> the five generator-side losses (:371-381).  `sd_g` / `sd_d`: reference-keyed
    parameter dicts; `hps`: dict with the data.* / train.* keys used below.  Returns (loss_disc, loss_gen_all, the six losses);
    the caller runs loss_gen_all.backward() (:383-387).  wav_aug = wav (the freeze_quantizer branch)."""
    from oracle import mel_ref
    h = hps
    spec = mel_ref.spectrogram(wav, h["filter_length"], h["hop_length"], h["win_length"])
    spec_lengths = wav_lengths // h["hop_length"]
    seg = h["segment_size"] // h["hop_length"]
    y_hat, kl_ssl, ids, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), _ = synthesizer_forward(
        sd_g, cfg, buffers, wav, wav, wav_lengths, spec, spec, spec_lengths, text, text_lengths, noise_p, noise_q, ids_slice,
        segment_size=seg, training=True)
    mel = mel_ref.spec_to_mel(spec, h["filter_length"], h["n_mel_channels"], h["sampling_rate"], h["mel_fmin"], h["mel_fmax"])
    idx = ids.view(-1, 1) + torch.arange(seg).view(1, -1)
    y_mel = torch.gather(mel, 2, idx.unsqueeze(1).expand(-1, mel.size(1), -1))
    y_hat_mel = mel_ref.mel_spectrogram(y_hat.squeeze(1), h["filter_length"], h["n_mel_channels"], h["sampling_rate"], h["hop_length"],
                                        h["win_length"], h["mel_fmin"], h["mel_fmax"])
    widx = (ids * h["hop_length"]).view(-1, 1) + torch.arange(h["segment_size"]).view(1, -1)
    y = torch.gather(wav, 1, widx).unsqueeze(1)
    dr, dg, _, _ = mpd_forward(sd_d, y, y_hat.detach())
    loss_disc = discriminator_loss(dr, dg)
    if d_update is not None:
        d_update(loss_disc)
    dr, dg, fr, fg = mpd_forward(sd_d, y, y_hat)
    loss_mel = F.l1_loss(y_mel, y_hat_mel) * h["c_mel"]
    loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * h["c_kl"]
    loss_fm = feature_loss(fr, fg)
    loss_gen = generator_loss(dg)
    loss_gen_all = loss_gen + loss_fm + loss_mel + kl_ssl + loss_kl
    return loss_disc, loss_gen_all, {"loss_disc": loss_disc, "loss_gen": loss_gen, "loss_fm": loss_fm, "loss_mel": loss_mel,
                                     "kl_ssl": kl_ssl, "loss_kl": loss_kl}
