"""Attention stacks of the VQ-VAE text / style encoders on the HIP kernels (csrc/attn_f32.hip, conv.hip, vqvae_ops.hip).

Mirrors ttts/vqvae/attentions.py (`Encoder` :10-88, `MultiHeadAttention` :177-375, `FFN` :377-432) and
`modules.LayerNorm` (ttts/vqvae/modules.py:19-31) with the reference's constructor arguments and state-dict keys.
Tensors stay in the reference's (B, C, T) layout end to end; a head is a contiguous block of d_k channel rows, so the
batched GEMMs address q/k/v in place by strides.
"""
import math

import torch
from torch import nn

from .. import ops
from .modules import Conv1d, add_scale, mul_mask


class _SeedSource:
    """Per-process dropout stream: every dropout site of a step draws a distinct 64-bit seed on the host (no device sync)."""
    counter = 0x5EED0000

    @classmethod
    def reseed(cls, seed, rank=0):
        """Fold the training seed and the data-parallel rank into the stream (replicas must not share dropout masks)."""
        cls.counter = 0x5EED0000 + ((int(seed) * 1000003 + int(rank) * 7919) & 0x3FFFFFFF) * 4096

    @classmethod
    def next(cls):
        cls.counter += 1
        return (cls.counter * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy, ctx.p, ctx.seed), None, None


def dropout(x, p, training):
    if not training or p == 0.0:
        return x
    return _DropoutFn.apply(x, float(p), _SeedSource.next())


class _LayerNormChFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mean, rstd = ops.layernorm_ch_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.refs = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .modules import _grad_slot
        x, gamma, mean, rstd = ctx.saved_tensors
        sg, sb = _grad_slot(ctx.refs[0]), _grad_slot(ctx.refs[1])
        if sg is not None and sb is not None:     # straight into the flat gradient arena: no zero fills, no autograd adds
            dx, _, _ = ops.layernorm_ch_bwd(dy, x, gamma, mean, rstd, dg=sg, db=sb)
            return dx, None, None, None
        dx, dg, db = ops.layernorm_ch_bwd(dy, x, gamma, mean, rstd)
        return dx, dg, db, None


class LayerNorm(nn.Module):
    """modules.LayerNorm(channels, eps): LayerNorm over the channel axis of (B, C, T)."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        return _LayerNormChFn.apply(x, self.gamma, self.beta, self.eps)


class _AttnFusedFn(torch.autograd.Function):
    """softmax(mask(scale q^T k)) v on (B, C, T) tensors without a window and without dropout: csrc/attn_cross.hip (three fused
    kernels, no [B, H, Tq, Tk] tensors).  The MRTE text<->audio cross-attention and the style encoder's self-attention."""

    @staticmethod
    def forward(ctx, q, k, v, qmask, kmask, H, scale, fill):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out, lse = ops.attn_cross_fwd(q, k, v, qmask, kmask, H, scale, fill)
        ctx.save_for_backward(q, k, v, qmask, kmask, out, lse)
        ctx.cfg = (H, scale, fill)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, qmask, kmask, out, lse = ctx.saved_tensors
        H, scale, fill = ctx.cfg
        dq, dk, dv = ops.attn_cross_bwd(q, k, v, qmask, kmask, out, dout.contiguous(), lse, H, scale, fill)
        return dq, dk, dv, None, None, None, None, None


def fused_attention_ok(C, H, window, p_drop):
    return (not window) and p_drop == 0 and C % H == 0 and (C // H) in (64, 96, 128)


class _AttnCoreFn(torch.autograd.Function):
    """softmax(mask(scale q^T k + rel_k)) (dropout) v + rel_v on (B, C, T) tensors; see csrc/attn_f32.hip."""

    @staticmethod
    def forward(ctx, q, k, v, ek, ev, qmask, kmask, H, window, scale, fill, p_drop, seed):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, C, Tq = q.shape
        Tk = k.shape[2]
        dk = C // H
        S = torch.empty(B, H, Tq, Tk, dtype=torch.float32, device=q.device)
        ops.bgemm(q, k, S, Tq, Tk, dk, (1, Tq), (Tk, 1), (Tk, 1), B, H, (C * Tq, dk * Tq), (C * Tk, dk * Tk),
                  (H * Tq * Tk, Tq * Tk), alpha=scale)
        P = ops.attn_softmax_fwd(S, q if window else None, ek if window else None, qmask, kmask, dk, window, scale, fill)
        Pd = ops.dropout(P, p_drop, seed) if p_drop > 0 else P
        out = torch.empty(B, C, Tq, dtype=torch.float32, device=q.device)
        ops.bgemm(v, Pd, out, dk, Tq, Tk, (Tk, 1), (1, Tk), (Tq, 1), B, H, (C * Tk, dk * Tk), (H * Tq * Tk, Tq * Tk),
                  (C * Tq, dk * Tq))
        if window:
            ops.attn_rel(Pd, out, ev, H, window, 1.0, 0)
        ctx.save_for_backward(q, k, v, ek, ev, qmask, kmask, P, Pd if p_drop > 0 else None)
        ctx.cfg = (H, window, scale, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, ek, ev, qmask, kmask, P, Pd = ctx.saved_tensors
        H, window, scale, p_drop, seed = ctx.cfg
        if Pd is None:
            Pd = P
        dout = dout.contiguous()
        B, C, Tq = q.shape
        Tk = k.shape[2]
        dk = C // H
        sP, sQ, sK = (H * Tq * Tk, Tq * Tk), (C * Tq, dk * Tq), (C * Tk, dk * Tk)
        dP = torch.empty_like(P)
        ops.bgemm(dout, v, dP, Tq, Tk, dk, (1, Tq), (Tk, 1), (Tk, 1), B, H, sQ, sK, sP)
        dev = dek = None
        if window:
            ops.attn_rel(dP, dout, ev, H, window, 1.0, 1)
            dev = torch.zeros_like(ev)
            ops.attn_rel(Pd, dout, dev, H, window, 1.0, 2)
        dv = torch.empty_like(v)
        ops.bgemm(dout, Pd, dv, dk, Tk, Tq, (Tq, 1), (Tk, 1), (Tk, 1), B, H, sQ, sP, sK)
        if p_drop > 0:
            dP = ops.dropout(dP, p_drop, seed)
        dS = ops.attn_softmax_bwd(dP, P, qmask, kmask)
        dq = torch.empty_like(q)
        ops.bgemm(k, dS, dq, dk, Tq, Tk, (Tk, 1), (1, Tk), (Tq, 1), B, H, sK, sP, sQ, alpha=scale)
        if window:
            ops.attn_rel(dS, dq, ek, H, window, scale, 0)
            dek = torch.zeros_like(ek)
            ops.attn_rel(dS, q, dek, H, window, scale, 2)
        dkk = torch.empty_like(k)
        ops.bgemm(q, dS, dkk, dk, Tk, Tq, (Tq, 1), (Tk, 1), (Tk, 1), B, H, sQ, sP, sK, alpha=scale)
        return dq, dkk, dv, dek, dev, None, None, None, None, None, None, None, None


def _split_outer_mask(attn_mask):
    """(B, 1, Tq, Tk) product mask -> its two factors (B, Tq), (B, Tk) (every mask on the path is q_mask x k_mask)."""
    m = attn_mask[:, 0]
    return (m.amax(dim=2) > 0).float().contiguous(), (m.amax(dim=1) > 0).float().contiguous()


class MultiHeadAttention(nn.Module):
    """attentions.MultiHeadAttention / vc_utils.MultiHeadAttention (identical classes).  `forward(x, c, attn_mask)` keeps
    the reference signature; `q_mask` / `k_mask` ((B, 1, T) factors of attn_mask) skip the factorisation."""

    def __init__(self, channels, out_channels, n_heads, p_dropout=0.0, window_size=None, heads_share=True,
                 block_length=None, proximal_bias=False, proximal_init=False):
        super().__init__()
        assert channels % n_heads == 0
        if block_length is not None or proximal_bias:
            raise NotImplementedError("block_length / proximal_bias are unused on the path")
        self.channels, self.out_channels, self.n_heads, self.p_dropout = channels, out_channels, n_heads, p_dropout
        self.window_size, self.heads_share = window_size, heads_share
        self.k_channels = channels // n_heads
        self.conv_q = Conv1d(channels, channels, 1)
        self.conv_k = Conv1d(channels, channels, 1)
        self.conv_v = Conv1d(channels, channels, 1)
        self.conv_o = Conv1d(channels, out_channels, 1)
        if window_size is not None:
            n_heads_rel = 1 if heads_share else n_heads
            rel_stddev = self.k_channels ** -0.5
            self.emb_rel_k = nn.Parameter(torch.randn(n_heads_rel, window_size * 2 + 1, self.k_channels) * rel_stddev)
            self.emb_rel_v = nn.Parameter(torch.randn(n_heads_rel, window_size * 2 + 1, self.k_channels) * rel_stddev)
        for c in (self.conv_q, self.conv_k, self.conv_v):
            nn.init.xavier_uniform_(c.weight)
        if proximal_init:
            with torch.no_grad():
                self.conv_k.weight.copy_(self.conv_q.weight)
                self.conv_k.bias.copy_(self.conv_q.bias)

    def forward(self, x, c, attn_mask=None, q_mask=None, k_mask=None, resid=None, bbias=None, omask=None):
        """`resid` (B, C, T), `bbias` (B, C) and `omask` (B, 1, T) are fused into the output projection:
        omask * (conv_o(att) + bbias + resid)."""
        q, k, v = self.conv_q(x), self.conv_k(c), self.conv_v(c)
        if q_mask is not None:
            qm, km = q_mask.reshape(x.shape[0], -1).contiguous(), k_mask.reshape(c.shape[0], -1).contiguous()
        elif attn_mask is not None:
            qm, km = _split_outer_mask(attn_mask)
        else:
            qm = km = None
        w = self.window_size or 0
        if w:
            assert x.shape[2] == c.shape[2], "Relative attention is only available for self-attention."
        p = self.p_dropout if self.training else 0.0
        if fused_attention_ok(q.shape[1], self.n_heads, w, p):      # the MRTE text<->audio cross-attention: fused kernels
            out = _AttnFusedFn.apply(q, k, v, qm, km, self.n_heads, 1.0 / math.sqrt(self.k_channels), -1e4)
        else:
            out = _AttnCoreFn.apply(q, k, v, self.emb_rel_k if w else None, self.emb_rel_v if w else None, qm, km, self.n_heads,
                                    w, 1.0 / math.sqrt(self.k_channels), -1e4, p, _SeedSource.next() if p > 0 else 0)
        return self.conv_o(out, resid=resid, bbias=bbias, omask=omask)


class FFN(nn.Module):
    """attentions.FFN (:377-432), non-causal, relu: conv_k(x*mask) -> relu -> drop -> conv_k(.*mask) -> *mask.
    relu and both mask multiplies are conv epilogues."""

    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0.0, activation=None, causal=False):
        super().__init__()
        if causal or activation == "gelu" or kernel_size % 2 == 0:
            raise NotImplementedError("only the non-causal relu FFN with odd kernels is on the path")
        self.in_channels, self.out_channels, self.filter_channels = in_channels, out_channels, filter_channels
        self.kernel_size, self.p_dropout = kernel_size, p_dropout
        self.conv_1 = Conv1d(in_channels, filter_channels, kernel_size, padding=(kernel_size - 1) // 2)
        self.conv_2 = Conv1d(filter_channels, out_channels, kernel_size, padding=(kernel_size - 1) // 2)

    def forward(self, x, x_mask):
        x = self.conv_1(mul_mask(x, x_mask), out_act="lrelu", out_slope=0.0, omask=x_mask)   # relu, then * mask
        x = dropout(x, self.p_dropout, self.training)
        return self.conv_2(x, omask=x_mask)


class Encoder(nn.Module):
    """attentions.Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, window_size)."""

    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0.0, window_size=4,
                 isflow=False, **kwargs):
        super().__init__()
        if isflow:
            raise NotImplementedError("isflow encoders are unused on the path")
        self.hidden_channels, self.filter_channels, self.n_heads, self.n_layers = hidden_channels, filter_channels, n_heads, n_layers
        self.kernel_size, self.p_dropout, self.window_size = kernel_size, p_dropout, window_size
        self.attn_layers = nn.ModuleList()
        self.norm_layers_1 = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.norm_layers_2 = nn.ModuleList()
        for _ in range(n_layers):
            self.attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads, p_dropout=p_dropout,
                                                       window_size=window_size))
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size, p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))

    def forward(self, x, x_mask, g=None):
        assert g is None
        x = mul_mask(x, x_mask)
        for i in range(self.n_layers):
            y = self.attn_layers[i](x, x, q_mask=x_mask, k_mask=x_mask)
            y = dropout(y, self.p_dropout, self.training)
            x = self.norm_layers_1[i](add_scale([x, y]))
            y = self.ffn_layers[i](x, x_mask)
            y = dropout(y, self.p_dropout, self.training)
            x = self.norm_layers_2[i](add_scale([x, y]))
        return mul_mask(x, x_mask)
