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
