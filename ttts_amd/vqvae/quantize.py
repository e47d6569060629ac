"""Residual vector quantizer with the reference's module surface -- `ResidualVectorQuantizer(dimension, n_q, bins, decay,
kmeans_init, kmeans_iters, threshold_ema_dead_code)`, `.forward(x, n_q=None, layers=None) -> (quantized, codes,
mean commit loss, quantized_list)`, `.encode`, `.decode`, and the state-dict keys
`vq.layers.{i}._codebook.{inited,cluster_size,embed,embed_avg}` (ttts/vqvae/quantize.py:28-119, core_vq.py:96-382) --
on the HIP kernels `ttts_vq_nearest_f32` (exact-fp32 nearest code, bit-identical indices), `ttts_vq_commit_f32`
(commitment MSE + its gradient) and `ttts_vq_ema_update_f32` (scatter-add EMA, no 4096 x 1024 one-hot).

The two RNG-driven, once-in-a-while paths -- k-means initialisation on the first batch (core_vq.py:71-93,141-148) and
dead-code replacement (core_vq.py:152-168) -- are host-orchestrated around the same kernels (nearest-code search for the
Lloyd assignment) and torch index ops; they are not part of the steady-state step.  Their only random input, the index
vector of `sample_vectors`, can be injected (`index_source`), which is how tests/golden/vq.npz `kmeans_*` pins them.
"""
import torch
import torch.nn as nn

from .. import ops


index_source = None     # tests: callable (n, num) -> LongTensor(num) replacing the random index draw of _sample_vectors


def _sample_vectors(samples, num):
    """core_vq.py:60-68: `num` rows of `samples`, without replacement when there are enough of them."""
    n = samples.shape[0]
    if index_source is not None:
        idx = index_source(n, num).to(samples.device)
    else:
        idx = torch.randperm(n, device=samples.device)[:num] if n >= num else torch.randint(0, n, (num,), device=samples.device)
    return samples[idx]


def _kmeans(samples, num_clusters, num_iters):
    samples = samples[:500]                                    # max_kmeans_samples (core_vq.py:73-74)
    means = _sample_vectors(samples, num_clusters)
    bins = None
    for _ in range(num_iters):
        buckets, _, _ = ops.vq_nearest(samples.contiguous(), means.contiguous(), want_xq=False)
        bins = torch.bincount(buckets, minlength=num_clusters)
        new_means = torch.zeros_like(means).index_add_(0, buckets, samples)
        new_means = new_means / bins.clamp_min(1)[:, None]
        means = torch.where((bins == 0)[:, None], means, new_means)
    return means, bins


class _QuantizeST(torch.autograd.Function):
    """nearest code + straight-through estimator + commitment loss, one autograd node."""

    @staticmethod
    def forward(ctx, x, embed, training, commitment_weight):
        flat = x.reshape(-1, x.shape[-1]).contiguous()
        idx, xq, _ = ops.vq_nearest(flat, embed)
        if training and commitment_weight > 0:
            loss = ops.vq_commit(flat, xq) * commitment_weight
        else:
            loss = torch.zeros((), dtype=x.dtype, device=x.device)
        ctx.save_for_backward(flat, xq)
        ctx.cfg = (training, commitment_weight, x.shape)
        ctx.mark_non_differentiable(idx)
        # training: the reference returns x + (q - x) (core_vq.py:311), which differs from q by a rounding
        out = flat + (xq - flat) if training else xq
        return out.view(x.shape), idx.view(x.shape[:-1]), loss

    @staticmethod
    def backward(ctx, g_q, g_idx, g_loss):
        flat, xq = ctx.saved_tensors
        training, w, shape = ctx.cfg
        dx = None
        if g_q is not None and training:
            dx = g_q.reshape(flat.shape).clone()               # quantize = x + (q - x).detach(): identity gradient
        if g_loss is not None and training and w > 0:
            dc = torch.zeros_like(flat)
            ops.vq_commit(flat, xq, dc, 1.0)                   # dc = 2 (x - q) / n
            dc = dc * (g_loss * w)
            dx = dc if dx is None else dx + dc
        return (dx.view(shape) if dx is not None else None), None, None, None


class EuclideanCodebook(nn.Module):
    def __init__(self, dim, codebook_size, kmeans_init=False, kmeans_iters=10, decay=0.99, epsilon=1e-5,
                 threshold_ema_dead_code=2):
        super().__init__()
        self.decay, self.codebook_size, self.kmeans_iters = decay, codebook_size, kmeans_iters
        self.epsilon, self.threshold_ema_dead_code = epsilon, threshold_ema_dead_code
        if kmeans_init:
            embed = torch.zeros(codebook_size, dim)
        else:
            embed = torch.empty(codebook_size, dim)
            nn.init.kaiming_uniform_(embed)
        self.register_buffer("inited", torch.Tensor([not kmeans_init]))
        self.register_buffer("cluster_size", torch.zeros(codebook_size))
        self.register_buffer("embed", embed)
        self.register_buffer("embed_avg", embed.clone())

    _inited_host = False       # host mirror of `inited` once it is known to be set (a device read per forward is a sync)

    @torch.no_grad()
    def init_embed_(self, data):
        if self._inited_host:
            return
        if bool(self.inited):
            self._inited_host = True
            return
        embed, cluster_size = _kmeans(data, self.codebook_size, self.kmeans_iters)
        self.embed.copy_(embed)
        self.embed_avg.copy_(embed)
        self.cluster_size.copy_(cluster_size)
        self.inited.fill_(1)
        self._inited_host = True

    def _load_from_state_dict(self, *a, **k):
        self._inited_host = False                          # a loaded checkpoint decides again
        return super()._load_from_state_dict(*a, **k)

    sync_free = False    # True (hipGraph capture / no host read-back in the step): skip the dead-code replacement -- see below

    @torch.no_grad()
    def expire_codes_(self, batch_samples):
        """core_vq.py:152-168.  NOTE (reference behaviour, kept): the rows this writes into `embed` are overwritten a few
        lines later in the same forward by `embed = embed_avg / smoothed cluster_size` (core_vq.py:225-228), so the
        replacement never survives a training step -- its only lasting effect is the random draw it consumes.  In
        `sync_free` mode (a captured step cannot read `any(expired)` back to the host) it is therefore skipped."""
        if self.threshold_ema_dead_code == 0 or self.sync_free:
            return
        expired = self.cluster_size < self.threshold_ema_dead_code
        if not bool(torch.any(expired)):
            return
        flat = batch_samples.reshape(-1, batch_samples.shape[-1])
        self.embed.copy_(torch.where(expired[:, None], _sample_vectors(flat, self.codebook_size), self.embed))

    def quantize(self, x):
        return ops.vq_nearest(x.contiguous(), self.embed, want_xq=False)[0]

    def dequantize(self, embed_ind):
        return nn.functional.embedding(embed_ind, self.embed)

    def encode(self, x):
        shape = x.shape
        return self.quantize(x.reshape(-1, shape[-1])).view(*shape[:-1])

    def decode(self, embed_ind):
        return self.dequantize(embed_ind)

    @torch.no_grad()
    def update_(self, flat, idx):
        """expire -> EMA(cluster_size, embed_avg) -> Laplace-normalise (core_vq.py:216-228), after the forward."""
        self.expire_codes_(flat)
        ops.vq_ema_update(flat, idx.reshape(-1), self.cluster_size, self.embed_avg, self.embed, self.decay, self.epsilon)


class VectorQuantization(nn.Module):
    def __init__(self, dim, codebook_size, codebook_dim=None, decay=0.99, epsilon=1e-5, kmeans_init=True,
                 kmeans_iters=50, threshold_ema_dead_code=2, commitment_weight=1.0):
        super().__init__()
        cd = dim if codebook_dim is None else codebook_dim
        self.project_in = nn.Linear(dim, cd) if cd != dim else nn.Identity()
        self.project_out = nn.Linear(cd, dim) if cd != dim else nn.Identity()
        self.epsilon, self.commitment_weight, self.codebook_size = epsilon, commitment_weight, codebook_size
        self._codebook = EuclideanCodebook(dim=cd, codebook_size=codebook_size, kmeans_init=kmeans_init,
                                           kmeans_iters=kmeans_iters, decay=decay, epsilon=epsilon,
                                           threshold_ema_dead_code=threshold_ema_dead_code)

    @property
    def codebook(self):
        return self._codebook.embed

    def encode(self, x):
        return self._codebook.encode(self.project_in(x.transpose(1, 2)))

    def decode(self, embed_ind):
        return self.project_out(self._codebook.decode(embed_ind)).transpose(1, 2)

    def forward(self, x):
        x = self.project_in(x.transpose(1, 2))                 # b d n -> b n d
        cb = self._codebook
        cb.init_embed_(x.detach().reshape(-1, x.shape[-1]))
        # the codebook the search sees is the one BEFORE this step's EMA update (core_vq.py:211-214 vs :216-228)
        quantize, embed_ind, loss = _QuantizeST.apply(x, cb.embed.clone() if self.training else cb.embed, self.training,
                                                      self.commitment_weight)
        if self.training:
            cb.update_(x.detach().reshape(-1, x.shape[-1]).contiguous(), embed_ind)
        quantize = self.project_out(quantize).transpose(1, 2)  # b n d -> b d n
        return quantize, embed_ind, loss.reshape(1)


class ResidualVectorQuantization(nn.Module):
    def __init__(self, *, num_quantizers, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([VectorQuantization(**kwargs) for _ in range(num_quantizers)])

    def forward(self, x, n_q=None, layers=None):
        quantized_out, residual = 0.0, x
        all_losses, all_indices, out_quantized = [], [], []
        n_q = n_q or len(self.layers)
        for i, layer in enumerate(self.layers[:n_q]):
            quantized, indices, loss = layer(residual)
            residual = residual - quantized
            quantized_out = quantized_out + quantized
            all_indices.append(indices)
            all_losses.append(loss)
            if layers and i in layers:
                out_quantized.append(quantized)
        return quantized_out, torch.stack(all_indices), torch.stack(all_losses), out_quantized

    def encode(self, x, n_q=None, st=None):
        residual, all_indices = x, []
        n_q, st = n_q or len(self.layers), st or 0
        for layer in self.layers[st:n_q]:
            indices = layer.encode(residual)
            residual = residual - layer.decode(indices)
            all_indices.append(indices)
        return torch.stack(all_indices)

    def decode(self, q_indices, st=0):
        out = torch.tensor(0.0, device=q_indices.device)
        for i, indices in enumerate(q_indices):
            out = out + self.layers[st + i].decode(indices)
        return out


class ResidualVectorQuantizer(nn.Module):
    def __init__(self, dimension=256, n_q=8, bins=1024, decay=0.99, kmeans_init=True, kmeans_iters=50,
                 threshold_ema_dead_code=2):
        super().__init__()
        self.n_q, self.dimension, self.bins, self.decay = n_q, dimension, bins, decay
        self.kmeans_init, self.kmeans_iters, self.threshold_ema_dead_code = kmeans_init, kmeans_iters, threshold_ema_dead_code
        self.vq = ResidualVectorQuantization(dim=dimension, codebook_size=bins, num_quantizers=n_q, decay=decay,
                                             kmeans_init=kmeans_init, kmeans_iters=kmeans_iters,
                                             threshold_ema_dead_code=threshold_ema_dead_code)

    def forward(self, x, n_q=None, layers=None):
        n_q = n_q if n_q else self.n_q
        if layers and max(layers) >= n_q:
            raise ValueError(f"Last layer index in layers: A {max(layers)}. Number of quantizers in RVQ: B {self.n_q}. "
                             "A must less than B.")
        quantized, codes, commit_loss, quantized_list = self.vq(x, n_q=n_q, layers=layers)
        return quantized, codes, torch.mean(commit_loss), quantized_list

    def encode(self, x, n_q=None, st=None):
        return self.vq.encode(x, n_q=n_q if n_q else self.n_q, st=st or 0)

    def decode(self, codes, st=0):
        return self.vq.decode(codes, st=st)
