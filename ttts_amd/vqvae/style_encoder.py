"""MelStyleEncoder (ttts/vqvae/modules.py:686-764 with LinearNorm :523-540, Mish :543-548, Conv1dGLU :551-570,
ConvNorm :573-602, MultiHeadAttention :606-661, ScaledDotProductAttention :664-683) on the HIP kernels.

The reference transposes to (B, T, C) for its Linear layers; here everything stays (B, C, T) and a Linear is a 1x1
convolution with the same (out, in) weight, so the state-dict keys and shapes are the reference's.
"""
import numpy as np
import torch
from torch import nn

from .. import ops
from .attentions import _AttnCoreFn, _SeedSource, dropout
from .modules import Conv1d, _Conv1dFn, add_scale, mul_mask


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        ctx.op = op
        ctx.save_for_backward(x)
        return ops.act_fwd(x, op)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd(dy, x, ctx.op), None


def mish(x):
    return _ActFn.apply(x, ops.ACT_MISH)


class _GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        ctx.kind = kind
        ctx.save_for_backward(x)
        return ops.gate_fwd(x, kind)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gate_bwd(dy, x, ctx.kind), None


class _MaskedMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        ctx.T = x.shape[2]
        ctx.save_for_backward(mask)
        return ops.masked_mean_fwd(x, mask)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return ops.masked_mean_bwd(dy, mask, ctx.T), None


class _Fc(nn.Module):
    """nn.Linear parameters ((out, in) weight) applied along the channel axis of (B, C, T)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        lin = nn.Linear(in_features, out_features, bias)      # same default init as the reference
        self.weight, self.bias = lin.weight, lin.bias

    def forward(self, x, resid=None):
        return _Conv1dFn.apply(x, self.weight.unsqueeze(-1), self.bias, resid, None, 1, 0, 1, 1.0, None)


class LinearNorm(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, spectral_norm=False):
        super().__init__()
        if spectral_norm:
            raise NotImplementedError("spectral_norm is unused on the path")
        self.fc = _Fc(in_channels, out_channels, bias)

    def forward(self, x):
        return self.fc(x)


class Mish(nn.Module):
    def forward(self, x):
        return mish(x)


class _Drop(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        return dropout(x, self.p, self.training)


class ConvNorm(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 spectral_norm=False):
        super().__init__()
        if spectral_norm:
            raise NotImplementedError("spectral_norm is unused on the path")
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = Conv1d(in_channels, out_channels, kernel_size, stride, padding=padding, dilation=dilation, bias=bias)

    def forward(self, x):
        return self.conv(x)


class Conv1dGLU(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, dropout):
        super().__init__()
        self.out_channels = out_channels
        self.conv1 = ConvNorm(in_channels, 2 * out_channels, kernel_size=kernel_size)
        self.p = dropout

    def forward(self, x):
        y = _GateFn.apply(self.conv1(x), ops.GATE_GLU)
        return add_scale([x, dropout(y, self.p, self.training)])


class StyleMultiHeadAttention(nn.Module):
    """modules.MultiHeadAttention(n_head, d_model, d_k, d_v, dropout) on (B, C, T): temperature sqrt(d_model), key padding
    mask filled with -inf, dropout on the probabilities and on the output, residual add."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.0, spectral_norm=False):
        super().__init__()
        if spectral_norm or d_k != d_v:
            raise NotImplementedError
        self.n_head, self.d_k, self.d_v, self.p = n_head, d_k, d_v, dropout
        self.w_qs = _Fc(d_model, n_head * d_k)
        self.w_ks = _Fc(d_model, n_head * d_k)
        self.w_vs = _Fc(d_model, n_head * d_v)
        self.temperature = float(np.power(d_model, 0.5))
        self.fc = _Fc(n_head * d_v, d_model)

    def forward(self, x, key_keep=None):
        """x (B, C, T); key_keep (B, T) float, 1 = attend / 0 = padded key (None: attend everywhere)."""
        q, k, v = self.w_qs(x), self.w_ks(x), self.w_vs(x)
        p = self.p if self.training else 0.0
        out = _AttnCoreFn.apply(q, k, v, None, None, None, key_keep, self.n_head, 0, 1.0 / self.temperature, -float("inf"),
                                p, _SeedSource.next() if p > 0 else 0)
        out = self.fc(out)
        return add_scale([dropout(out, self.p, self.training), x])


class MelStyleEncoder(nn.Module):
    """MelStyleEncoder(n_mel_channels, style_hidden, style_vector_dim, style_kernel_size, style_head, dropout);
    forward(x (B, n_mel, T), mask (B, 1, T) with 1 = valid) -> (B, style_vector_dim, 1)."""

    def __init__(self, n_mel_channels=80, style_hidden=128, style_vector_dim=256, style_kernel_size=5, style_head=2,
                 dropout=0.1):
        super().__init__()
        self.in_dim, self.hidden_dim, self.out_dim = n_mel_channels, style_hidden, style_vector_dim
        self.kernel_size, self.n_head, self.dropout = style_kernel_size, style_head, dropout
        self.spectral = nn.Sequential(LinearNorm(self.in_dim, self.hidden_dim), Mish(), _Drop(dropout),
                                      LinearNorm(self.hidden_dim, self.hidden_dim), Mish(), _Drop(dropout))
        self.temporal = nn.Sequential(Conv1dGLU(self.hidden_dim, self.hidden_dim, self.kernel_size, dropout),
                                      Conv1dGLU(self.hidden_dim, self.hidden_dim, self.kernel_size, dropout))
        self.slf_attn = StyleMultiHeadAttention(self.n_head, self.hidden_dim, self.hidden_dim // self.n_head,
                                                self.hidden_dim // self.n_head, dropout)
        self.fc = LinearNorm(self.hidden_dim, self.out_dim)

    def forward(self, x, mask=None):
        keep = mask.reshape(x.shape[0], -1).contiguous().float() if mask is not None else None
        x = self.spectral(x)
        x = self.temporal(x)
        if keep is not None:
            x = mul_mask(x, keep)                    # masked_fill(pad, 0)
        x = self.slf_attn(x, keep)
        x = self.fc(x)
        w = _MaskedMeanFn.apply(x, keep)             # temporal average over the valid frames
        return w.unsqueeze(-1)
