"""Losses of the VQ-VAE-GAN step on the HIP reductions (csrc/losses.hip) -- same names, arguments and return structure
as ttts/vqvae/losses.py:7-61, plus `l1_loss` for train.py:389.

Differences, all host-side: every value is a device tensor (the reference's `discriminator_loss` pulls 12 Python floats
per step with `.item()`; here `r_losses` / `g_losses` hold 0-dim tensors), sums are deterministic two-stage reductions.
"""
import torch

from .. import ops


def _pair_dense(a, b):
    """Same-order dense storage of two same-layout tensors without copying: the discriminators hand out (B, C, H, W)
    feature maps as permuted views of their (B*W, C, H) working layout; a mean over all elements does not care."""
    if a.is_contiguous() and b.is_contiguous():
        return a, b
    if a.dim() == 4 and a.stride() == b.stride():
        pa, pb = a.permute(0, 3, 1, 2), b.permute(0, 3, 1, 2)
        if pa.is_contiguous() and pb.is_contiguous():
            return pa, pb
    return a.contiguous(), b.contiguous()


class _ReduceLoss(torch.autograd.Function):
    """scale * sum term(a, b) as a 0-dim tensor; gradient to b (ABSDIFF) or a (squares)."""

    @staticmethod
    def forward(ctx, a, b, mode, scale):
        ctx.mode, ctx.scale = mode, scale
        ctx.save_for_backward(a, b)
        return ops.reduce_loss(a, b, mode, scale).view(())

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        g = gout.contiguous().view(1).float()
        d = ops.reduce_loss_bwd(a, b, ctx.mode, ctx.scale, g)
        if ctx.mode == ops.RED_ABSDIFF:
            return None, d, None, None
        return d, None, None, None


def _aligned(t):
    """The reduction kernels read 16 bytes per lane: a view that starts mid-allocation (the generated half of a batched
    discriminator output) is copied once; whole tensors pass through."""
    return t if t is None or t.data_ptr() % 16 == 0 else t.clone()


def _mean_term(a, b, mode, mult=1.0):
    return _ReduceLoss.apply(_aligned(a), _aligned(b), mode, mult / a.numel())


def l1_loss(target, pred):
    """F.l1_loss(target, pred) with the gradient flowing to `pred` only (train.py:389: y_mel is data)."""
    t, p = _pair_dense(target.detach().float(), pred.float())
    return _mean_term(t, p, ops.RED_ABSDIFF)


class _SumOfTerms(torch.autograd.Function):
    """sum_i scale_i * sum term(a_i, b_i) accumulated into ONE device scalar (`reduce_loss(..., accumulate=True)`), gradient to every
    b_i.  The feature loss is 36 such terms: as 36 autograd nodes it also cost 36 scalar `add` launches, 36 AddBackward nodes and a
    final `* 2`; here the factor is folded into the scales.  inputs = (a_0, b_0, a_1, b_1, ...)."""

    @staticmethod
    def forward(ctx, scales, *ab):
        ctx.scales = scales
        ctx.save_for_backward(*ab)
        out = torch.zeros(1, dtype=torch.float32, device=ab[0].device)
        for i, sc in enumerate(scales):
            ops.reduce_loss(ab[2 * i], ab[2 * i + 1], ops.RED_ABSDIFF, sc, out=out, accumulate=True)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        ab = ctx.saved_tensors
        g = gout.contiguous().view(1).float()
        grads = [None]
        for i, sc in enumerate(ctx.scales):
            grads += [None, ops.reduce_loss_bwd(ab[2 * i], ab[2 * i + 1], ops.RED_ABSDIFF, sc, g)]
        return tuple(grads)


def feature_loss(fmap_r, fmap_g):
    terms, scales = [], []
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            r, g = _pair_dense(rl.float().detach(), gl.float())
            terms += [_aligned(r), _aligned(g)]
            scales.append(2.0 / r.numel())
    return _SumOfTerms.apply(tuple(scales), *terms)


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    loss = 0
    r_losses, g_losses = [], []
    for dr, dg in zip(disc_real_outputs, disc_generated_outputs):
        r_loss = _mean_term(dr.float().contiguous(), None, ops.RED_SQ_ONE_MINUS)
        g_loss = _mean_term(dg.float().contiguous(), None, ops.RED_SQ)
        loss = loss + (r_loss + g_loss)
        r_losses.append(r_loss.detach())
        g_losses.append(g_loss.detach())
    return loss, r_losses, g_losses


def generator_loss(disc_outputs):
    loss = 0
    gen_losses = []
    for dg in disc_outputs:
        l = _mean_term(dg.float().contiguous(), None, ops.RED_SQ_ONE_MINUS)
        gen_losses.append(l)
        loss = loss + l
    return loss, gen_losses


class _KlLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_p, logs_q, m_p, logs_p, z_mask):
        out = ops.kl_loss_fwd(z_p, logs_q, m_p, logs_p, z_mask)
        ctx.save_for_backward(z_p, logs_q, m_p, logs_p, z_mask, out)
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        z_p, logs_q, m_p, logs_p, z_mask, out = ctx.saved_tensors
        dz, dlq, dm, dlp = ops.kl_loss_bwd(z_p, logs_q, m_p, logs_p, z_mask, out, gout.contiguous().view(1).float())
        return dz, dlq, dm, dlp, None


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    """z_p, logs_q, m_p, logs_p: [b, h, t]; z_mask [b, 1, t]  (losses.py:45-61)."""
    return _KlLoss.apply(z_p.float(), logs_q.float(), m_p.float(), logs_p.float(), z_mask.float())
