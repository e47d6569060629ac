"""The training side of the reference's Gaussian diffusion (ttts/utils/diffusion.py): `get_named_beta_schedule` (:83-107),
`space_timesteps` (:1223-1272), `GaussianDiffusion.__init__ / q_sample` (:200-260) and `SpacedDiffusion.training_losses`
(:930-1014, :1181-1220) for the configuration the trainer uses (ttts/diffusion/train.py:92-94): epsilon prediction,
learned-range variance, mse loss + variational-bound term.  Coefficient tables are built in float64 numpy exactly as the
reference does; the per-element arithmetic runs in the `q_sample` / `diffusion_loss` HIP kernels.

Not built: sampling loops (p_sample / ddim / dpm++), the KL-only and x0 / previous-x parametrisations.
"""
import math

import numpy as np
import torch

from .. import ops


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2   # noqa: E731
        n = num_diffusion_timesteps
        return np.array([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)])
    raise NotImplementedError("unknown beta schedule: %s" % schedule_name)


def space_timesteps(num_timesteps, section_counts):
    """diffusion.py:1223-1272."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError("cannot create exactly %d steps with an integer stride" % num_timesteps)
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start, all_steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError("cannot divide section of %d steps into %d" % (size, count))
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur, taken = 0.0, []
        for _ in range(count):
            taken.append(start + round(cur))
            cur += frac
        all_steps += taken
        start += size
    return set(all_steps)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_out, x_start, x_t, noise, t, table):
        model_out = model_out.contiguous()
        terms, loss = ops.diffusion_loss_fwd(model_out, x_start, x_t, noise, t, table)
        ctx.save_for_backward(model_out, x_start, x_t, noise, t, table)
        ctx.mark_non_differentiable(terms)
        return loss.view(()), terms

    @staticmethod
    def backward(ctx, g, _g_terms):
        model_out, x_start, x_t, noise, t, table = ctx.saved_tensors
        return ops.diffusion_loss_bwd(model_out, x_start, x_t, noise, t, table, g.reshape(1).contiguous()), None, None, None, None, None


class SpacedDiffusion:
    """SpacedDiffusion(use_timesteps, model_mean_type='epsilon', model_var_type='learned_range', loss_type='mse', betas, ...)."""

    def __init__(self, use_timesteps, *, betas, model_mean_type="epsilon", model_var_type="learned_range", loss_type="mse",
                 rescale_timesteps=False, conditioning_free=False, conditioning_free_k=1, ramp_conditioning_free=True,
                 sampler="ddim"):
        if (model_mean_type, model_var_type, loss_type) != ("epsilon", "learned_range", "mse"):
            raise NotImplementedError("only epsilon / learned_range / mse (the trainer's configuration) is built")
        self.use_timesteps = set(use_timesteps)
        base = np.array(betas, dtype=np.float64)
        self.original_num_steps = len(base)
        base_ac = np.cumprod(1.0 - base, axis=0)
        last, new_betas, self.timestep_map = 1.0, [], []
        for i, a in enumerate(base_ac):                       # diffusion.py:1186-1194
            if i in self.use_timesteps:
                new_betas.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        betas = np.array(new_betas, dtype=np.float64)
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        self.rescale_timesteps = rescale_timesteps
        self.conditioning_free, self.conditioning_free_k = conditioning_free, conditioning_free_k
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self._table_host = np.stack([self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, self.sqrt_recip_alphas_cumprod,
                                     self.sqrt_recipm1_alphas_cumprod, self.posterior_mean_coef1, self.posterior_mean_coef2,
                                     self.posterior_log_variance_clipped, np.log(betas)], axis=1).astype(np.float32)
        self._tables = {}
        self._map = {}

    def table(self, device):
        key = str(device)
        if key not in self._tables:
            self._tables[key] = torch.from_numpy(self._table_host).to(device).contiguous()
            self._map[key] = torch.tensor(self.timestep_map, dtype=torch.int64, device=device)
        return self._tables[key]

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        return ops.q_sample(x_start, noise, t.long().contiguous(), self.table(x_start.device))

    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """Returns {"loss", "mse", "vb"} of shape (N,) like the reference, plus "loss_mean" (scalar, differentiable): the
        trainer's `["loss"].mean()` without a separate reduction (train.py:174-182)."""
        model_kwargs = model_kwargs or {}
        if noise is None:
            noise = torch.randn_like(x_start)
        t = t.long().contiguous()
        table = self.table(x_start.device)
        x_t = ops.q_sample(x_start, noise, t, table)
        new_ts = self._map[str(x_start.device)][t]                       # _WrappedModel (:1275-1287)
        if self.rescale_timesteps:
            new_ts = (new_ts.float() * (1000.0 / self.original_num_steps)).long()
        model_out = model(x_t, new_ts, **model_kwargs)
        B, C = x_t.shape[:2]
        if model_out.shape != (B, C * 2, *x_t.shape[2:]):
            raise ValueError("learned-range variance needs a (B, 2C, T) model output, got %s" % (tuple(model_out.shape),))
        loss_mean, terms = _LossFn.apply(model_out, x_start.contiguous(), x_t, noise.contiguous(), t, table)
        return {"loss": terms[:, 2], "mse": terms[:, 0], "vb": terms[:, 1], "loss_mean": loss_mean}
