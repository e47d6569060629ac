"""`AA_diffusion` with the reference's constructor / forward signature and state-dict surface (ttts/diffusion/aa_model.py:
182-287; 283 tensors at the shipped config), its blocks (`ResBlock` :70-131, `DiffusionLayer` :134-148, `RefEncoder` :150-177),
and `AttentionBlock` / `GroupNorm32` / `normalization` / `RelativePositionBias` of ttts/utils/utils.py:104-215 and
ttts/utils/xtransformers.py:146-185 -- every tensor op a HIP kernel behind the C ABI (conv family, fp32 attention pieces,
csrc/diffusion_ops.hip); torch only moves memory (cat / slice / views).

Built: the training path (`forward(x, timesteps, latent, refer)`, 1-D, `use_scale_shift_norm=True`, relative position
embeddings).  `conditioning_free=True` is supported (inference-time guidance branch); fp16 flags are accepted and ignored
(fp32 with split-bf16 matrix-core convolutions, as the VQ-VAE path).
The two random choices of the reference forward (`unconditioned_percentage` mask, `layer_drop`) use the host RNG like the
reference (`torch.rand` / `random.random`) and can be injected (`uncond=`, `drop_layers=`) for parity tests.
"""
import math
import os
import random

import torch
import torch.nn as nn

from .. import ops
from ..vqvae.attentions import MultiHeadAttention
from ..vqvae.modules import Conv1d, _grad_slot
from ..vqvae.style_encoder import _ActFn

TACOTRON_MEL_MAX = 5.5451774444795624753378569716654
TACOTRON_MEL_MIN = -16.118095650958319788125940182791


def denormalize_tacotron_mel(norm_mel):
    return norm_mel / 0.18215


def normalize_tacotron_mel(mel):
    """aa_model.py:21-23 (clamp from below at -TACOTRON_MEL_MAX, then scale)."""
    return torch.clamp(mel, min=-TACOTRON_MEL_MAX) * 0.18215


def timestep_embedding(timesteps, dim, max_period=10000):
    """aa_model.py:32-51; timesteps (N,) int64 on the GPU."""
    return ops.timestep_embedding(timesteps.long().contiguous(), dim, max_period)


def silu(x):
    return _ActFn.apply(x, ops.ACT_SILU)


# ---- GroupNorm32 ------------------------------------------------------------------------------------------------------------
class _GroupNormFn(torch.autograd.Function):
    """y = act(GroupNorm(x) [* (1 + scale) + shift]); ss (B, 2C) = scale | shift or None."""

    @staticmethod
    def forward(ctx, x, gamma, beta, ss, groups, act_silu):
        x = x.contiguous()
        ss = ss.contiguous() if ss is not None else None
        y, mean, rstd = ops.groupnorm_fwd(x, gamma, beta, groups, ss, act_silu)
        ctx.save_for_backward(x, gamma, beta, ss, mean, rstd)
        ctx.cfg = (groups, act_silu)
        ctx.refs = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, ss, mean, rstd = ctx.saved_tensors
        groups, act_silu = ctx.cfg
        sg, sb = _grad_slot(ctx.refs[0]), _grad_slot(ctx.refs[1])
        if sg is not None and sb is not None:
            dx, _, _, dss = ops.groupnorm_bwd(dy, x, gamma, beta, ss, mean, rstd, groups, act_silu, dgamma=sg, dbeta=sb)
            return dx, None, None, dss, None, None
        dx, dg, db, dss = ops.groupnorm_bwd(dy, x, gamma, beta, ss, mean, rstd, groups, act_silu)
        return dx, dg, db, dss, None, None


class GroupNorm32(nn.Module):
    """nn.GroupNorm(groups, channels) computing in fp32 (utils.py:113-115).  `forward(x, scale_shift=None, silu=False)` takes the
    fused neighbours of the ResBlock (timestep scale / shift, the SiLU that follows)."""

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def forward(self, x, scale_shift=None, silu=False):
        return _GroupNormFn.apply(x, self.weight, self.bias, scale_shift, self.num_groups, bool(silu))


def normalization(channels):
    """utils.py:118-133."""
    groups = 32
    if channels <= 16:
        groups = 8
    elif channels <= 64:
        groups = 16
    while channels % groups != 0:
        groups = int(groups / 2)
    assert groups > 2
    return GroupNorm32(groups, channels)


# ---- precision of the 1 x 1 convolutions / linear layers ---------------------------------------------------------------------------
# "f32" (default): fp32 on split-bf16 matrix-core products, as everywhere else on the path.  "fp8": BASELINE config #5's "fp8 MFMA
# GEMMs" -- operands quantised per tensor to OCP e4m3 (current scaling), v_mfma_f32_32x32x16_fp8_fp8 with fp32 accumulation, for the
# forward, the data gradient and the weight gradient (csrc/fp8_gemm.hip; oracle/fp8_ref.py).  Activations between the layers stay
# fp32 (wider than the config's bf16).  Switch: set_precision() or TTTS_DIFFUSION_PRECISION.
_PRECISION = {"mode": os.environ.get("TTTS_DIFFUSION_PRECISION", "f32")}
if _PRECISION["mode"] not in ("f32", "fp8"):          # a typo must not silently select the default arithmetic
    raise ValueError("TTTS_DIFFUSION_PRECISION must be 'f32' or 'fp8' (got %r)" % _PRECISION["mode"])


def set_precision(mode):
    if mode not in ("f32", "fp8"):
        raise ValueError("diffusion precision must be 'f32' or 'fp8'")
    prev = _PRECISION["mode"]
    _PRECISION["mode"] = mode
    return prev


class _Conv1x1Fp8Fn(torch.autograd.Function):
    """y = W x (+ bias) (+ resid) on the fp8 matrix cores; x (B, Cin, T), w (Cout, Cin, 1).  The backward works from the forward's
    e4m3 codes (the activation row-wise, the weight in both layouts): nothing is measured or quantised twice."""

    @staticmethod
    def forward(ctx, x, w, bias, resid):
        x = x.contiguous()
        y, xq, wq = ops.conv1x1_fp8_fwd(x, w, bias, resid.contiguous() if resid is not None else None)
        ctx.saved = (xq, wq)
        ctx.cin = x.shape[1]
        ctx.refs = (w, bias)
        ctx.has = (bias is not None, resid is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xq, wq = ctx.saved
        ctx.saved = None
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        dw = db = None
        out = None
        if need[1]:
            slot = _grad_slot(ctx.refs[0])
            out = slot if slot is not None else torch.zeros_like(ctx.refs[0])
            dw = None if slot is not None else out
        dx = ops.conv1x1_fp8_bwd(dy, xq, wq, ctx.cin, need_dx=need[0], dw_out=out)
        if ctx.has[0] and need[2]:
            bslot = _grad_slot(ctx.refs[1])
            db = ops.conv1d_bias_grad(dy, out=bslot)
            if bslot is not None:
                db = None
        return dx, dw, db, (dy if ctx.has[1] and need[3] else None)


class _LinearFp8Fn(torch.autograd.Function):
    """y = x W^T + bias on the fp8 matrix cores; x (R, Cin), w (Cout, Cin)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x = x.contiguous()
        R, Cin = x.shape
        Cout = w.shape[0]
        ax, aw = ops.fp8_amax(x), ops.fp8_amax(w)
        xq, wq = ops.fp8_quant(x, ax), ops.fp8_quant(w, aw)
        y = torch.empty(R, Cout, dtype=torch.float32, device=x.device)
        ops.fp8_gemm_nt(wq, xq, y, aw, ax, Cout, R, xq.shape[1], bias=bias, y_strides=(0, 1, Cout))
        ctx.save_for_backward(x, w, ax, aw)
        ctx.refs = (w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, ax, aw = ctx.saved_tensors
        dy = dy.contiguous()
        R, Cin = x.shape
        Cout = w.shape[0]
        need = ctx.needs_input_grad
        dx = dw = db = None
        ady = ops.fp8_amax(dy)
        if need[0]:
            wtq = ops.fp8_quant_transpose(w.view(1, Cout, Cin), aw)[0]          # (Cin, Coutp)
            dyq = ops.fp8_quant(dy, ady)                                        # (R, Coutp)
            dx = torch.empty(R, Cin, dtype=torch.float32, device=x.device)
            ops.fp8_gemm_nt(wtq, dyq, dx, aw, ady, Cin, R, dyq.shape[1], y_strides=(0, 1, Cin))
        if need[1]:
            slot = _grad_slot(ctx.refs[0])
            out = slot if slot is not None else torch.zeros_like(w)
            dytq = ops.fp8_quant_transpose(dy.view(1, R, Cout), ady)[0]         # (Cout, Rp)
            xtq = ops.fp8_quant_transpose(x.view(1, R, Cin), ax)[0]             # (Cin, Rp)
            ops.fp8_gemm_nt(dytq, xtq, out, ady, ax, Cout, Cin, dytq.shape[1], y_strides=(0, Cin, 1), accumulate=True)
            dw = None if slot is not None else out
        if need[2]:
            bslot = _grad_slot(ctx.refs[1])
            db = ops.conv1d_bias_grad(dy.t().contiguous().view(1, Cout, R), out=bslot)
            if bslot is not None:
                db = None
        return dx, dw, db


class Conv1x1(Conv1d):
    """A Conv1d of kernel size 1 that follows the diffusion precision switch (same parameters / state-dict keys as Conv1d)."""

    def forward(self, x, resid=None, **kw):
        if _PRECISION["mode"] == "fp8" and self.kernel_size == 1 and not kw:
            return _Conv1x1Fp8Fn.apply(x, self.weight, self.bias, resid)
        return super().forward(x, resid=resid, **kw)


class Linear(nn.Module):
    """nn.Linear (2-D weight, default init) evaluated as a 1x1 convolution over a length-1 signal."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        ref = nn.Linear(in_features, out_features)
        self.weight, self.bias = nn.Parameter(ref.weight.detach().clone()), nn.Parameter(ref.bias.detach().clone())

    def forward(self, x):
        from ..vqvae.modules import _Conv1dFn
        if _PRECISION["mode"] == "fp8":
            return _LinearFp8Fn.apply(x.reshape(-1, self.in_features), self.weight, self.bias).reshape(*x.shape[:-1], self.out_features)
        y = _Conv1dFn.apply(x.reshape(-1, self.in_features, 1), self.weight.unsqueeze(-1), self.bias, None, None, 1, 0, 1, 1.0, None)
        return y.reshape(*x.shape[:-1], self.out_features)


class SiLU(nn.Module):
    def forward(self, x):
        return silu(x)


# ---- attention with T5-bucket relative position bias -----------------------------------------------------------------------------
_BUCKETS = {}


def _bucket_table(max_len, num_buckets, max_distance, device):
    """int32 [2 * off + 1]: bucket of relative position r = j - i at index r + off, computed with the reference's own formula
    (RelativePositionBias._relative_position_bucket, causal=False; xtransformers.py:155-174) on the host, once per length."""
    off = 1
    while off < max_len:
        off *= 2
    key = (off, num_buckets, max_distance, str(device))
    if key not in _BUCKETS:
        rel = torch.arange(-off, off + 1)
        n = -rel
        nb = num_buckets // 2
        ret = (n < 0).long() * nb
        n = torch.abs(n)
        max_exact = nb // 2
        is_small = n < max_exact
        val_if_large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
        val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, nb - 1))
        _BUCKETS[key] = (ret + torch.where(is_small, n, val_if_large)).to(torch.int32).to(device)
    return _BUCKETS[key]


class _RelPosBiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, bucket, H, Tq, Tk, scale):
        ctx.save_for_backward(bucket)
        ctx.cfg = (table.shape[0], scale)
        ctx.ref = table
        return ops.relpos_bias_fwd(table.contiguous(), bucket, H, Tq, Tk, scale)

    @staticmethod
    def backward(ctx, dbias):
        (bucket,) = ctx.saved_tensors
        nb, scale = ctx.cfg
        slot = _grad_slot(ctx.ref)
        d = ops.relpos_bias_bwd(dbias.contiguous().unsqueeze(0), bucket, nb, scale, out=slot)
        return (None if slot is not None else d), None, None, None, None, None


class _Embedding(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, d))


class RelativePositionBias(nn.Module):
    """xtransformers.py:146-185 (non-causal).  `bias(i, j)` returns the (H, i, j) additive logits (already * scale)."""

    def __init__(self, scale, causal=False, num_buckets=32, max_distance=128, heads=8):
        super().__init__()
        if causal:
            raise NotImplementedError("causal buckets are not on the path")
        self.scale, self.num_buckets, self.max_distance, self.heads = scale, num_buckets, max_distance, heads
        self.relative_attention_bias = _Embedding(num_buckets, heads)

    def bias(self, i, j):
        w = self.relative_attention_bias.weight
        return _RelPosBiasFn.apply(w, _bucket_table(max(i, j), self.num_buckets, self.max_distance, w.device), self.heads, i, j,
                                   float(self.scale))


class AttentionBlock(nn.Module):
    """utils.py:172-215 (1-D, relative position embeddings; mask unused on the path)."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, do_checkpoint=True, relative_pos_embeddings=False):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.norm = normalization(channels)
        self.qkv = Conv1x1(channels, channels * 3, 1)
        self.proj_out = Conv1x1(channels, channels, 1)
        with torch.no_grad():                                           # zero_module (utils.py:104-110)
            self.proj_out.weight.zero_(); self.proj_out.bias.zero_()
        self.relative_pos_embeddings = RelativePositionBias(scale=(channels // self.num_heads) ** .5, causal=False, heads=num_heads,
                                                            num_buckets=32, max_distance=64) if relative_pos_embeddings else None

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("attention masks are not used on the diffusion path")
        T = x.shape[-1]
        qkv = self.qkv(self.norm(x))
        if self.relative_pos_embeddings is not None:
            products = 1 if _PRECISION["mode"] == "fp8" else 3
            fused = (_FUSED_ATTN and self.channels // self.num_heads == 32 and T <= ops.attn_relpos_max_t(products))
            h = (_AttnRelPosFused if fused else _AttnWithBias).apply(
                qkv, self.relative_pos_embeddings.relative_attention_bias.weight, _bucket_table(T, 32, 64, x.device),
                self.num_heads, float(self.relative_pos_embeddings.scale))
        else:
            raise NotImplementedError("every AttentionBlock on the diffusion path uses relative position embeddings")
        return self.proj_out(h, resid=x)


_FUSED_ATTN = os.environ.get("TTTS_DIFFUSION_FUSED_ATTN", "1") == "1"    # (A/B switch: 0 = the materialised-scores path below)


class _AttnRelPosFused(torch.autograd.Function):
    """AttentionBlock's attention as one forward and three backward launches (csrc/attn_relpos.hip): scores, probabilities and the
    (H, T, T) bias never exist in memory; saved for the backward: qkv, the output and one log-sum-exp per (b, h, query).
    Operand arithmetic follows the step's precision mode: split-bf16 (fp32-equivalent) by default, plain bf16 -- the reference's
    autocast attention (ttts/diffusion/train.py:171) -- in the "fp8" mode."""

    @staticmethod
    def forward(ctx, qkv, table, bucket, H, scale):
        qkv = qkv.contiguous()
        products = 1 if _PRECISION["mode"] == "fp8" else 3
        tab = table.contiguous()
        out, lse = ops.attn_relpos_fwd(qkv, tab, bucket, H, scale, products)
        ctx.save_for_backward(qkv, tab, bucket, out, lse)
        ctx.cfg = (H, scale, products)
        ctx.ref = table
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, tab, bucket, out, lse = ctx.saved_tensors
        H, scale, products = ctx.cfg
        need = ctx.needs_input_grad[1]
        slot = _grad_slot(ctx.ref) if need else None
        dqkv, dtable = ops.attn_relpos_bwd(qkv, tab, bucket, out, dout.contiguous(), lse, H, scale, products, dtable=slot,
                                           need_dtable=need)
        return dqkv, (None if (slot is not None or not need) else dtable), None, None, None


class _AttnWithBias(torch.autograd.Function):
    """Legacy attention + bucket bias in one autograd node: the bias gradient goes from dS straight to the (buckets, H) table."""

    @staticmethod
    def forward(ctx, qkv, table, bucket, H, scale):
        qkv = qkv.contiguous()
        B, W, T = qkv.shape
        ch = W // (3 * H)
        bias = ops.relpos_bias_fwd(table.contiguous(), bucket, H, T, T, scale)
        v5 = qkv.view(B, H, 3, ch, T)
        q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
        sQ = (W * T, 3 * ch * T)
        S = torch.empty(B, H, T, T, dtype=torch.float32, device=qkv.device)
        ops.bgemm(q, k, S, T, T, ch, (1, T), (T, 1), (T, 1), B, H, sQ, sQ, (H * T * T, T * T), alpha=1.0 / math.sqrt(ch))
        P = ops.softmax_bias_fwd(S, bias)
        out = torch.empty(B, H * ch, T, dtype=torch.float32, device=qkv.device)
        ops.bgemm(v, P, out, ch, T, T, (T, 1), (1, T), (T, 1), B, H, sQ, (H * T * T, T * T), (H * ch * T, ch * T))
        ctx.save_for_backward(qkv, P, bucket)
        ctx.cfg = (H, scale, table.shape[0])
        ctx.ref = table
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, P, bucket = ctx.saved_tensors
        H, scale, nb = ctx.cfg
        dout = dout.contiguous()
        B, W, T = qkv.shape
        ch = W // (3 * H)
        v5 = qkv.view(B, H, 3, ch, T)
        q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(B, H, 3, ch, T)
        dq, dk, dv = d5[:, :, 0], d5[:, :, 1], d5[:, :, 2]
        sQ, sP, sO = (W * T, 3 * ch * T), (H * T * T, T * T), (H * ch * T, ch * T)
        a = 1.0 / math.sqrt(ch)
        dP = torch.empty_like(P)
        ops.bgemm(dout, v, dP, T, T, ch, (1, T), (T, 1), (T, 1), B, H, sO, sQ, sP)
        ops.bgemm(dout, P, dv, ch, T, T, (T, 1), (T, 1), (T, 1), B, H, sO, sP, sQ)
        dS = ops.attn_softmax_bwd(dP, P, None, None)
        ops.bgemm(k, dS, dq, ch, T, T, (T, 1), (1, T), (T, 1), B, H, sQ, sP, sQ, alpha=a)
        ops.bgemm(q, dS, dk, ch, T, T, (T, 1), (T, 1), (T, 1), B, H, sQ, sP, sQ, alpha=a)
        dtable = None
        if ctx.needs_input_grad[1]:
            slot = _grad_slot(ctx.ref)
            dtable = ops.relpos_bias_bwd(dS, bucket, nb, scale, out=slot)
            if slot is not None:
                dtable = None
        return dqkv, dtable, None, None, None


# ---- blocks ---------------------------------------------------------------------------------------------------------------------
class _Seq(nn.Module):
    """Container giving children the reference's numeric names (nn.Sequential indices) without running them in sequence."""

    def __init__(self, **mods):
        super().__init__()
        for k, m in mods.items():
            self.add_module(k, m)

    def __getitem__(self, i):
        return getattr(self, str(i))


class TimestepBlock(nn.Module):
    pass


class ResBlock(TimestepBlock):
    """aa_model.py:70-131 (dims 1, efficient_config: 1x1 input conv; scale-shift norm)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, dims=2, kernel_size=3, efficient_config=True,
                 use_scale_shift_norm=False):
        super().__init__()
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_scale_shift_norm = use_scale_shift_norm
        if not use_scale_shift_norm:
            raise NotImplementedError("only the scale-shift ResBlock variant is on the diffusion path")
        if dropout != 0:
            raise NotImplementedError("dropout inside the diffusion ResBlock is 0 in the shipped config and not built")
        padding = {1: 0, 3: 1, 5: 2}[kernel_size]
        eff_kernel, eff_padding = (1, 0) if efficient_config else (3, 1)
        self.in_layers = _Seq(**{"0": normalization(channels), "1": SiLU(),
                                 "2": (Conv1x1 if eff_kernel == 1 else Conv1d)(channels, self.out_channels, eff_kernel, padding=eff_padding)})
        self.emb_layers = _Seq(**{"0": SiLU(), "1": Linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels)})
        self.out_layers = _Seq(**{"0": normalization(self.out_channels), "1": SiLU(), "2": nn.Identity(),
                                  "3": Conv1d(self.out_channels, self.out_channels, kernel_size, padding=padding)})
        self.skip_connection = nn.Identity() if self.out_channels == channels else (Conv1x1 if eff_kernel == 1 else Conv1d)(
            channels, self.out_channels, eff_kernel, padding=eff_padding)

    def forward(self, x, emb):
        h = self.in_layers[2](self.in_layers[0](x, silu=True))
        emb_out = self.emb_layers[1](silu(emb))
        h = self.out_layers[0](h, scale_shift=emb_out, silu=True)           # norm * (1 + scale) + shift, then SiLU: one kernel
        skip = x if isinstance(self.skip_connection, nn.Identity) else self.skip_connection(x)
        return self.out_layers[3](h, resid=skip)


class DiffusionLayer(TimestepBlock):
    """aa_model.py:134-148."""

    def __init__(self, model_channels, dropout, num_heads):
        super().__init__()
        self.resblk = ResBlock(model_channels, model_channels, dropout, model_channels, dims=1, use_scale_shift_norm=True)
        self.attn = AttentionBlock(model_channels, num_heads, relative_pos_embeddings=True)

    def forward(self, x, time_emb, refer=None):
        y = self.resblk(x, time_emb)
        if refer is not None:
            y = torch.cat([y, refer], dim=-1)
        y = self.attn(y)
        if refer is not None:
            y = y[:, :, :-refer.shape[-1]]
        return y


class _MeanLastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, C, T = x.shape
        ctx.T = T
        mask = torch.ones(B, T, dtype=torch.float32, device=x.device)
        ctx.save_for_backward(mask)
        return ops.masked_mean_fwd(x.contiguous(), mask)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return ops.masked_mean_bwd(dy.contiguous(), mask, ctx.T)


class RefEncoder(nn.Module):
    """aa_model.py:150-177."""

    def __init__(self, ref_dim, dim, num_latents=32, num_heads=8):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(num_latents, ref_dim) * 0.02)
        self.cross_attention = MultiHeadAttention(ref_dim, ref_dim, num_heads)
        self.enc = _Seq(**{"0": Conv1d(ref_dim, dim, 3, padding=1),
                           **{str(i): AttentionBlock(dim, num_heads, relative_pos_embeddings=True) for i in range(1, 5)}})

    def forward(self, x):
        batch = x.shape[0]
        latents = self.latents.t().unsqueeze(0).expand(batch, -1, -1).contiguous()       # "n d -> b d n"
        latents = self.cross_attention(latents, x)
        h = self.enc[0](torch.cat((latents, x), -1))
        for i in range(1, 5):
            h = self.enc[i](h)
        # the reference slices the CHANNEL axis with latents.shape[1] (= ref_dim, a no-op) and averages over every position
        return _MeanLastFn.apply(h[:, :self.latents.shape[1], :])


class _SelectRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, use, a, vec):
        ctx.save_for_backward(use)
        ctx.ref = vec
        return ops.select_rows_fwd(use, a, vec.reshape(-1).contiguous())

    @staticmethod
    def backward(ctx, dout):
        (use,) = ctx.saved_tensors
        da, dvec = ops.select_rows_bwd(use, dout, need_da=ctx.needs_input_grad[1])
        return None, da, dvec.view_as(ctx.ref) if ctx.needs_input_grad[2] else None


class _InterpNearestFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Tout):
        ctx.Tin = x.shape[-1]
        return ops.interp_nearest_fwd(x, Tout)

    @staticmethod
    def backward(ctx, dy):
        return ops.interp_nearest_bwd(dy, ctx.Tin), None


class AA_diffusion(nn.Module):
    def __init__(self, model_channels=512, num_layers=8, in_channels=100, in_latent_channels=512, out_channels=200, dropout=0,
                 num_heads=16, use_fp16=False, layer_drop=.1, unconditioned_percentage=.1):
        super().__init__()
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.dropout, self.num_heads = dropout, num_heads
        self.unconditioned_percentage, self.enable_fp16, self.layer_drop = unconditioned_percentage, use_fp16, layer_drop
        C, H = model_channels, num_heads
        ab = lambda: AttentionBlock(C, H, relative_pos_embeddings=True)   # noqa: E731
        # registration order = the reference's (aa_model.py:206-236): the state dict / parameter order depends on it
        self.inp_block = Conv1d(in_channels, C, 3, 1, 1)
        self.time_embed = _Seq(**{"0": Linear(C, C), "1": SiLU(), "2": Linear(C, C)})
        self.code_norm = normalization(C)
        self.latent_conditioner = _Seq(**{"0": Conv1d(in_latent_channels, C, 3, padding=1), "1": ab(), "2": ab(), "3": ab()})
        self.unconditioned_embedding = nn.Parameter(torch.randn(1, C, 1))
        self.conditioning_timestep_integrator = _Seq(**{str(i): DiffusionLayer(C, dropout, H) for i in range(3)})
        self.refer_enc = _Seq(**{"0": Conv1d(in_channels, C, 3, padding=1), "1": ab(), "2": ab(), "3": ab(), "4": RefEncoder(C, C)})
        self.integrating_conv = Conv1x1(C * 2, C, kernel_size=1)
        self.layers = nn.ModuleList([DiffusionLayer(C, dropout, H) for _ in range(num_layers)] +
                                    [ResBlock(C, C, dropout, dims=1, use_scale_shift_norm=True) for _ in range(3)])
        self.out = _Seq(**{"0": normalization(C), "1": SiLU(), "2": Conv1d(C, out_channels, 3, padding=1)})

    def timestep_independent(self, latent, refer, expected_seq_len, uncond=None):
        h = self.latent_conditioner[0](latent)
        for i in range(1, 4):
            h = self.latent_conditioner[i](h)
        r = self.refer_enc[0](refer)
        for i in range(1, 4):
            r = self.refer_enc[i](r)
        r = self.refer_enc[4](r)                                               # (B, C)
        B, C = r.shape
        # code_norm(latent_emb) + refer_emb[..., None]: the per-sample shift of the fused GroupNorm kernel (scale = 0)
        ss = torch.cat([torch.zeros_like(r), r], dim=1)
        latent_emb = self.code_norm(h, scale_shift=ss)
        if uncond is None and self.training and self.unconditioned_percentage > 0:
            uncond = torch.rand((B,), device=latent_emb.device) < self.unconditioned_percentage
        if uncond is not None:
            latent_emb = _SelectRowsFn.apply(uncond.to(torch.uint8).contiguous(), latent_emb, self.unconditioned_embedding)
        return _InterpNearestFn.apply(latent_emb, expected_seq_len)

    def forward(self, x, timesteps, latent=None, refer=None, conditioning_free=False, uncond=None, drop_layers=None):
        if conditioning_free:
            B, T = x.shape[0], x.shape[-1]
            use = torch.ones(B, dtype=torch.uint8, device=x.device)
            latent_emb = _SelectRowsFn.apply(use, torch.zeros(B, self.model_channels, T, device=x.device), self.unconditioned_embedding)
        else:
            latent_emb = self.timestep_independent(latent, refer, x.shape[-1], uncond)
        te = timestep_embedding(timesteps, self.model_channels)
        time_emb = self.time_embed[2](silu(self.time_embed[0](te)))
        for i in range(3):
            latent_emb = self.conditioning_timestep_integrator[i](latent_emb, time_emb)
        h = self.inp_block(x)
        h = self.integrating_conv(torch.cat([h, latent_emb], dim=1))
        n = len(self.layers)
        for i, lyr in enumerate(self.layers):
            if drop_layers is not None:
                skip = i in drop_layers and i != 0 and i != n - 1
            else:
                skip = self.training and self.layer_drop > 0 and i != 0 and i != n - 1 and random.random() < self.layer_drop
            if not skip:
                h = lyr(h, time_emb)
        return self.out[2](self.out[0](h, silu=True))
