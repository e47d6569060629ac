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
def det_fill(name, shape):
    """Deterministic, construction-order-independent parameter fill shared by tools/make_goldens.py, the oracle tests and
    the GPU parity tests (so multi-million-parameter state dicts need not be stored)."""
    import zlib
    import numpy as np
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    a = rng.standard_normal(size=tuple(shape), dtype=np.float32)
    if name.endswith("weight_g") or name.endswith("original0"):
        a = np.float32(1.0) + np.float32(0.1) * np.abs(a)            # ~ ||v|| = 1.2 * sqrt(fan_in)/sqrt(fan_in) scale below
    elif len(shape) > 1:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        a = a * np.float32(1.2 / fan_in ** 0.5)
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
