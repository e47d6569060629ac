"""Convolutional building blocks of the VQ-VAE-GAN path on the HIP conv family (csrc/conv.hip).

Mirrors, with the reference's constructor arguments and state-dict keys:
  * `Conv1d` / `ConvTranspose1d` (+ both weight-norm styles: `torch.nn.utils.weight_norm` -> `weight_g`/`weight_v`
    (ttts/vqvae/vq2.py:10,364) and `torch.nn.utils.parametrizations.weight_norm` -> `parametrizations.weight.original0/1`
    (ttts/vqvae/modules.py:8))
  * `ResBlock1`   ttts/vqvae/modules.py:224-318
  * `Generator`   ttts/vqvae/vq2.py:341-415 (HiFi-GAN decoder)

Every convolution, its leaky-relu prologue, bias, residual add and tanh epilogue, the weight-norm reparametrisation and
all of their gradients run in `libttts_hip.so`; torch provides autograd bookkeeping and storage only.  There is no CPU
path: the ops raise `TttsError` off-GPU.
"""
import math

import torch
from torch import nn

from .. import ops

LRELU_SLOPE = 0.1   # ttts/vqvae/modules.py:16


def get_padding(kernel_size, dilation=1):
    """ttts/utils/commons.py:12-13."""
    return int((kernel_size * dilation - dilation) / 2)


# ---- autograd glue ------------------------------------------------------------------------------------------------------
class _WeightNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g):
        w, norm = ops.weight_norm_fwd(v, g)
        ctx.save_for_backward(v, g, norm)
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g, norm = ctx.saved_tensors
        dv, dg = ops.weight_norm_bwd(dw, v, g, norm)
        return dv, dg


class _Conv1dFn(torch.autograd.Function):
    """y = act(bias + bbias + conv1d(lrelu(x, in_slope), w) + resid);  out_act None | 'tanh' | 'lrelu'."""

    @staticmethod
    def forward(ctx, x, w, bias, resid, bbias, stride, pad, dil, in_slope, out_act, groups=1, out_slope=LRELU_SLOPE):
        y = ops.conv1d_fwd(x, w, bias, resid, stride, pad, dil, in_slope, out_act, bbias=bbias, groups=groups,
                           out_slope=out_slope)
        ctx.save_for_backward(x, w, y if out_act else None)
        ctx.cfg = (stride, pad, dil, in_slope, out_act, groups, out_slope, bias is not None, resid is not None,
                   bbias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, dil, in_slope, out_act, groups, out_slope, has_b, has_r, has_bb = ctx.cfg
        dy = dy.contiguous()
        if out_act == "tanh":
            dy = ops.tanh_bwd(dy, y)
        elif out_act == "lrelu":
            dy = ops.lrelu_bwd(dy, y, out_slope)
        need = ctx.needs_input_grad
        dx = dw = db = dbb = None
        if need[0]:
            gate = x if in_slope != 1.0 else None
            dx = ops.conv1d_dgrad(dy, w, x.shape[2], stride, pad, dil, gate=gate, gate_slope=in_slope, groups=groups)
        if need[1]:
            dw = ops.conv1d_wgrad(dy, x, w.shape[2], stride, pad, dil, x_slope=in_slope, groups=groups)
        if has_b and need[2]:
            db = ops.conv1d_bias_grad(dy)
        if has_bb and need[4]:
            B, C, L = dy.shape
            dbb = ops.conv1d_bias_grad(dy.view(1, B * C, L)).view(B, C)
        return dx, dw, db, (dy if has_r and need[3] else None), dbb, None, None, None, None, None, None, None


class _ConvTranspose1dFn(torch.autograd.Function):
    """y = bias + conv_transpose1d(lrelu(x, in_slope), w);  w (Cin, Cout, K)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, in_slope):
        lout = (x.shape[2] - 1) * stride - 2 * pad + w.shape[2]
        y = ops.conv1d_dgrad(x, w, lout, stride, pad, 1, bias=bias, in_slope=in_slope)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, in_slope, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, in_slope, has_b = ctx.cfg
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        dx = dw = db = None
        if need[0]:
            dx = ops.conv1d_fwd(dy, w, stride=stride, pad=pad, gate=x if in_slope != 1.0 else None, gate_slope=in_slope)
        if need[1]:
            dw = ops.conv1d_wgrad(x, dy, w.shape[2], stride, pad, 1, dy_slope=in_slope)
        if has_b and need[2]:
            db = ops.conv1d_bias_grad(dy)
        return dx, dw, db, None, None, None


class _AddScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, *xs):
        ctx.scale, ctx.n = scale, len(xs)
        return ops.add_scale(xs, scale)

    @staticmethod
    def backward(ctx, dy):
        g = ops.add_scale([dy], ctx.scale) if ctx.scale != 1.0 else dy
        return (None,) + (g,) * ctx.n


def add_scale(xs, scale=1.0):
    return _AddScaleFn.apply(scale, *xs)


# ---- modules ------------------------------------------------------------------------------------------------------------
class _Originals(nn.Module):
    """Holder giving the `parametrizations.weight.original0/original1` keys of the new-style weight norm."""

    def __init__(self, g, v):
        super().__init__()
        self.original0 = nn.Parameter(g)
        self.original1 = nn.Parameter(v)


class _ConvBase(nn.Module):
    def _init_params(self, wshape, fan_in, bias_ch, bias):
        w = torch.empty(wshape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))           # torch's Conv default (reset_parameters)
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            self.bias = nn.Parameter(torch.empty(bias_ch).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._wn = None

    def apply_weight_norm(self, style):
        """style 'old': torch.nn.utils.weight_norm (weight_g, weight_v); 'new': parametrizations.weight_norm."""
        w = self.weight.detach()
        g = w.flatten(1).norm(dim=1).view(-1, *([1] * (w.dim() - 1))).clone()
        del self.weight
        if style == "old":
            self.weight_g = nn.Parameter(g)
            self.weight_v = nn.Parameter(w.clone())
        else:
            self.parametrizations = nn.ModuleDict({"weight": _Originals(g, w.clone())})
        self._wn = style
        return self

    def _g(self):
        return self.weight_g if self._wn == "old" else self.parametrizations["weight"].original0

    def _v(self):
        return self.weight_v if self._wn == "old" else self.parametrizations["weight"].original1

    def effective_weight(self):
        if self._wn is None:
            return self.weight
        return _WeightNormFn.apply(self._v(), self._g())


class Conv1d(_ConvBase):
    """nn.Conv1d(in, out, k, stride, padding, dilation, bias) (groups = 1) on the HIP kernels.  `forward` takes the fused
    neighbours: `in_slope` (leaky-relu on the input), `resid` (added to the output), `bbias` (per-sample bias (B, C)),
    `out_act` ('tanh' | 'lrelu' on the output)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        self._init_params((out_channels, in_channels // groups, kernel_size), in_channels // groups * kernel_size,
                          out_channels, bias)

    def forward(self, x, in_slope=1.0, resid=None, bbias=None, out_act=None, out_slope=LRELU_SLOPE):
        return _Conv1dFn.apply(x, self.effective_weight(), self.bias, resid, bbias, self.stride, self.padding, self.dilation,
                               float(in_slope), out_act, self.groups, float(out_slope))


class Conv2dK1(_ConvBase):
    """nn.Conv2d(in, out, (k, 1), (stride, 1), padding=(pad, 0)) -- the only Conv2d shape on the path (DiscriminatorP,
    vq2.py:425-472).  Parameters keep the 4-D reference shapes ((out, in, k, 1), weight_g (out, 1, 1, 1)); the module
    computes on (B*W, C, H) tensors, i.e. as a 1-D convolution along H with the period axis folded into the batch."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding = stride, padding
        self._init_params((out_channels, in_channels, kernel_size, 1), in_channels * kernel_size, out_channels, bias)

    def forward(self, x, in_slope=1.0, out_act=None, out_slope=LRELU_SLOPE):
        return _Conv1dFn.apply(x, self.effective_weight().squeeze(-1), self.bias, None, None, self.stride, self.padding, 1,
                               float(in_slope), out_act, 1, float(out_slope))


class ConvTranspose1d(_ConvBase):
    """nn.ConvTranspose1d(in, out, k, stride, padding) (output_padding 0, dilation 1, groups 1)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding = stride, padding
        self._init_params((in_channels, out_channels, kernel_size), out_channels * kernel_size, out_channels, bias)

    def forward(self, x, in_slope=1.0):
        return _ConvTranspose1dFn.apply(x, self.effective_weight(), self.bias, self.stride, self.padding, float(in_slope))


def _wn_conv_new(channels, k, d):
    # the reference's `convs.apply(init_weights)` runs AFTER weight_norm and writes `.weight.data.normal_` into the
    # recomputed temporary, i.e. it changes no parameter: the effective init is torch's Conv default with g = ||v||
    return Conv1d(channels, channels, k, 1, dilation=d, padding=get_padding(k, d)).apply_weight_norm("new")


class ResBlock1(nn.Module):
    """ttts/vqvae/modules.py:224-318: x <- x + c2(lrelu(c1(lrelu(x)))) for three (dilated, plain) conv pairs.  The two
    leaky-relus and the residual add are fused into the convolutions."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([_wn_conv_new(channels, kernel_size, d) for d in dilation])
        self.convs2 = nn.ModuleList([_wn_conv_new(channels, kernel_size, 1) for _ in dilation])

    def forward(self, x, x_mask=None):
        if x_mask is not None:
            raise NotImplementedError("ResBlock1 with x_mask is not on the training path (vq2.py never passes one)")
        for c1, c2 in zip(self.convs1, self.convs2):
            xt = c1(x, in_slope=LRELU_SLOPE)
            x = c2(xt, in_slope=LRELU_SLOPE, resid=x)
        return x


class Generator(nn.Module):
    """HiFi-GAN decoder, ttts/vqvae/vq2.py:341-415 (resblock type "1")."""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super().__init__()
        if str(resblock) != "1":
            raise NotImplementedError("only ResBlock1 (vqvae/config.json: resblock '1') is built")
        self.num_kernels = len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_rates)
        self.conv_pre = Conv1d(initial_channel, upsample_initial_channel, 7, 1, padding=3)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            up = ConvTranspose1d(upsample_initial_channel // (2 ** i), upsample_initial_channel // (2 ** (i + 1)), k, u,
                                 padding=(k - u) // 2)
            self.ups.append(up.apply_weight_norm("old"))    # `self.ups.apply(init_weights)` is a no-op, see _wn_conv_new
        self.resblocks = nn.ModuleList()
        ch = upsample_initial_channel
        for i in range(len(self.ups)):
            ch = upsample_initial_channel // (2 ** (i + 1))
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(ResBlock1(ch, k, d))
        self.conv_post = Conv1d(ch, 1, 7, 1, padding=3, bias=False)
        if gin_channels != 0:
            self.cond = Conv1d(gin_channels, upsample_initial_channel, 1)

    def forward(self, x, g=None):
        bbias = None
        if g is not None:
            bbias = self.cond(g).squeeze(-1)               # (B, C, 1) broadcast over time -> per-sample bias of conv_pre
        x = self.conv_pre(x, bbias=bbias)
        for i in range(self.num_upsamples):
            x = self.ups[i](x, in_slope=LRELU_SLOPE)
            xs = [self.resblocks[i * self.num_kernels + j](x) for j in range(self.num_kernels)]
            x = add_scale(xs, 1.0 / self.num_kernels)
        return self.conv_post(x, in_slope=0.01, out_act="tanh")   # F.leaky_relu default slope (vq2.py:404)
