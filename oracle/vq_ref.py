"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference VQ codebook path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Restates (reference = /root/reference, adelacvg/ttts):
* `EuclideanCodebook.quantize`      ttts/vqvae/core_vq.py:174-182   (L2 argmin, ties -> lowest index)
* `EuclideanCodebook.dequantize`    ttts/vqvae/core_vq.py:187-189
* `EuclideanCodebook.forward`       ttts/vqvae/core_vq.py:205-230   (train: expire -> EMA -> Laplace-normalise)
* `ema_inplace`, `laplace_smoothing` ttts/vqvae/core_vq.py:47-52
* `VectorQuantization.forward`      ttts/vqvae/core_vq.py:303-322   (straight-through, commitment MSE)
* `ResidualVectorQuantization.forward` / `ResidualVectorQuantizer.forward`
                                    ttts/vqvae/core_vq.py:336-359, ttts/vqvae/quantize.py:70-94 (n_q = 1 on the path)

* `sample_vectors`, `kmeans`, `init_embed_`, `expire_codes_` / `replace_`   ttts/vqvae/core_vq.py:60-93,141-168

Parity pin: fixture `tests/golden/vq.npz` generated from the imported reference by
`tools/make_goldens.py` (the reference has no tests for this path).  k-means initialisation and dead-code
replacement draw index vectors from torch's global RNG (`randperm(n)[:num]` / `randint(0, n, (num,))`); the
fixture cases `kmeans_*` inject those index vectors (`draws`), everything after the draw is pinned.
"""
import numpy as np
import torch
import torch.nn.functional as F


def quantize(x, embed):
    """x (N, D) fp32, embed (K, D) fp32 -> int64 (N).  Expression order is the reference's:
    -( (sum x^2  -  2 * x @ E^T)  +  sum E^2 ), then first-max."""
    e = embed.t()
    dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ e + e.pow(2).sum(0, keepdim=True))
    return dist.max(dim=-1).indices


def near_tie_audit(x, embed, idx, ulps=4.0):
    """Rows whose fp64 best/second-best distance gap is within `ulps` fp32 ulps of the distance
    magnitude: on those rows any fp32 summation order may legitimately pick either code.
    Returns a bool mask (N)."""
    xd, ed = x.double(), embed.double()
    d = (xd * xd).sum(1, keepdim=True) - 2 * xd @ ed.t() + (ed * ed).sum(1)[None]
    top2 = torch.topk(-d, 2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]).abs()
    mag = d.abs().max(dim=1).values.clamp_min(1e-30)
    return gap <= ulps * mag * float(np.finfo(np.float32).eps)


def kmeans(samples, num_clusters, num_iters, draw):
    """core_vq.py:71-93: at most 500 samples, means = samples[draw] (the injected `sample_vectors` index vector), then
    `num_iters` Lloyd iterations with the DIRECT squared distance (not the expanded form of `quantize`), empty clusters
    keep their previous mean.  Returns (means (K, D), bins (K) int64 of the last iteration)."""
    samples = samples[:500]
    means = samples[draw]
    bins = None
    for _ in range(num_iters):
        diffs = samples[:, None, :] - means[None, :, :]
        buckets = (-(diffs ** 2).sum(-1)).max(dim=-1).indices
        bins = torch.bincount(buckets, minlength=num_clusters)
        new_means = torch.zeros_like(means).index_add_(0, buckets, samples) / bins.clamp_min(1)[:, None]
        means = torch.where((bins == 0)[:, None], means, new_means)
    return means, bins


def codebook_forward(x, buffers, training, decay=0.99, epsilon=1e-5, threshold_ema_dead_code=2, kmeans_iters=10, draws=None):
    """EuclideanCodebook.forward.  `buffers` = dict(embed (K,D), embed_avg (K,D), cluster_size (K)[, inited (1)]),
    updated IN PLACE when training.  x (..., D).  Returns (quantize (..., D), embed_ind (...)).
    `draws`: list of injected `sample_vectors` index vectors, consumed in call order (k-means init, then expiry)."""
    shape = x.shape
    flat = x.reshape(-1, shape[-1])
    draws = list(draws or [])
    if "inited" in buffers and not bool(buffers["inited"]):                       # init_embed_ (core_vq.py:141-150)
        means, bins = kmeans(flat, buffers["embed"].shape[0], kmeans_iters, draws.pop(0))
        buffers["embed"].copy_(means); buffers["embed_avg"].copy_(means)
        buffers["cluster_size"].copy_(bins.to(buffers["cluster_size"].dtype)); buffers["inited"].fill_(1)
    embed = buffers["embed"]
    K = embed.shape[0]
    ind = quantize(flat, embed)
    onehot = F.one_hot(ind, K).type(x.dtype)
    q = F.embedding(ind.view(*shape[:-1]), embed)
    if training:
        expired = buffers["cluster_size"] < threshold_ema_dead_code
        if threshold_ema_dead_code and bool(torch.any(expired)):                   # expire_codes_ / replace_ (:152-168)
            if not draws:
                raise NotImplementedError("dead-code replacement draws from the global RNG: inject `draws`")
            buffers["embed"].copy_(torch.where(expired[:, None], flat[draws.pop(0)], buffers["embed"]))
            # (the normalisation below overwrites `embed` from `embed_avg`, so the replaced rows do not survive the step --
            #  a property of the reference, kept)
        buffers["cluster_size"].mul_(decay).add_(onehot.sum(0), alpha=1 - decay)
        embed_sum = flat.t() @ onehot
        buffers["embed_avg"].mul_(decay).add_(embed_sum.t(), alpha=1 - decay)
        cs = buffers["cluster_size"]
        smoothed = (cs + epsilon) / (cs.sum() + K * epsilon) * cs.sum()
        buffers["embed"].copy_(buffers["embed_avg"] / smoothed.unsqueeze(1))
    return q, ind.view(*shape[:-1])


def vq_forward(x_bdn, buffers, training, commitment_weight=1.0, **kw):
    """VectorQuantization.forward.  x (B, D, N) -> (quantize (B, D, N), ind (B, N), loss (1,))."""
    x = x_bdn.transpose(1, 2)
    q, ind = codebook_forward(x.detach(), buffers, training, **kw)
    loss = torch.zeros(1, dtype=x.dtype, device=x.device)
    if training:
        q = x + (q - x).detach()
        if commitment_weight > 0:
            loss = loss + F.mse_loss(q.detach(), x) * commitment_weight
    return q.transpose(1, 2), ind, loss


def rvq_forward(x_bdn, buffers, training, **kw):
    """ResidualVectorQuantizer.forward with n_q = 1, layers=[0] (vq2.py:835,851-852):
    -> (quantized, codes (1, B, N), mean commit loss (), [quantized])."""
    q, ind, loss = vq_forward(x_bdn, buffers, training, **kw)
    return q, ind.unsqueeze(0), loss.mean(), [q]
