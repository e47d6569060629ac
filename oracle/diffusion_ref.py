"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's diffusion mel-denoiser train step
(BASELINE config #5).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Restates, as plain functions over a state dict (reference = /root/reference, adelacvg/ttts):
* `AA_diffusion.forward / timestep_independent`, `DiffusionLayer`, `ResBlock` (scale-shift), `RefEncoder`, `timestep_embedding`
                                                                  ttts/diffusion/aa_model.py:32-287
* `AttentionBlock`, `QKVAttentionLegacy`, `GroupNorm32` / `normalization`   ttts/utils/utils.py:113-215
* `RelativePositionBias` (T5 buckets, non-causal, 32 buckets, max distance 64)  ttts/utils/xtransformers.py:146-185
* `MultiHeadAttention` (cross-attention, no window)               ttts/utils/vc_utils.py:514-627
* `GaussianDiffusion.__init__ / q_sample / q_posterior_mean_variance / p_mean_variance (epsilon, learned_range) /
  _vb_terms_bpd / training_losses (mse)`, `normal_kl`, `discretized_gaussian_log_likelihood`
                                                                  ttts/utils/diffusion.py:17-82,162-282,284-395,903-1014
* step body of `Trainer.train` (AdamW(1e-4, (0.9, 0.999), wd 0.01), clip 1.0, LambdaLR warm-up 1000)
                                                                  ttts/diffusion/train.py:69-73,119-120,156-200

Randomness of the reference forward (`unconditioned_percentage` mask, `layer_drop`) is INJECTED: `uncond` (B,) bool and
`drop_layers` (set of layer indices) are arguments.

Parity pin: `tests/golden/diffusion.npz`, produced by `tools/make_goldens.py diffusion` from the imported reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---- deterministic parameters -------------------------------------------------------------------------------------------------
def det_fill(name, shape, gain=1.0):
    """Construction-order-independent parameter fill shared by tools/make_goldens.py, the oracle tests and the GPU parity tests
    (the reference zero-initialises every attention output projection, which would hide the attention path)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    a = rng.standard_normal(size=tuple(shape), dtype=np.float32)
    if len(shape) == 1:
        a = np.float32(1.0) + np.float32(0.1) * a if name.endswith("weight") else np.float32(0.05) * a   # norm gains / biases
    elif name.endswith("relative_attention_bias.weight") or name.endswith("latents") or name == "unconditioned_embedding":
        a = a * np.float32(0.3)
    else:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        a = a * np.float32(gain / fan_in ** 0.5)
    return torch.from_numpy(a)


def attention_block_spec(c, heads):
    return [("norm.weight", (c,)), ("norm.bias", (c,)), ("qkv.weight", (3 * c, c, 1)), ("qkv.bias", (3 * c,)),
            ("proj_out.weight", (c, c, 1)), ("proj_out.bias", (c,)),
            ("relative_pos_embeddings.relative_attention_bias.weight", (32, heads))]


def res_block_spec(c, emb):
    return [("in_layers.0.weight", (c,)), ("in_layers.0.bias", (c,)), ("in_layers.2.weight", (c, c, 1)), ("in_layers.2.bias", (c,)),
            ("emb_layers.1.weight", (2 * c, emb)), ("emb_layers.1.bias", (2 * c,)), ("out_layers.0.weight", (c,)),
            ("out_layers.0.bias", (c,)), ("out_layers.3.weight", (c, c, 3)), ("out_layers.3.bias", (c,))]


def param_spec(cfg):
    """Names and shapes of AA_diffusion(**cfg).named_parameters() in registration order (aa_model.py:182-236); every
    state-dict entry of the reference is a parameter."""
    C, H, L = cfg["model_channels"], cfg["num_heads"], cfg["num_layers"]
    cin, clat, cout = cfg["in_channels"], cfg["in_latent_channels"], cfg["out_channels"]
    spec = []
    add = lambda pfx, items: spec.extend((pfx + k, s) for k, s in items)   # noqa: E731
    conv = lambda o, i, k: [("weight", (o, i, k)), ("bias", (o,))]          # noqa: E731
    dl = lambda: [("resblk." + k, s) for k, s in res_block_spec(C, C)] + [("attn." + k, s) for k, s in attention_block_spec(C, H)]  # noqa: E731
    spec.append(("unconditioned_embedding", (1, C, 1)))
    add("inp_block.", conv(C, cin, 3))
    add("time_embed.0.", [("weight", (C, C)), ("bias", (C,))])
    add("time_embed.2.", [("weight", (C, C)), ("bias", (C,))])
    add("code_norm.", [("weight", (C,)), ("bias", (C,))])
    add("latent_conditioner.0.", conv(C, clat, 3))
    for i in range(1, 4):
        add("latent_conditioner.%d." % i, attention_block_spec(C, H))
    for i in range(3):
        add("conditioning_timestep_integrator.%d." % i, dl())
    add("refer_enc.0.", conv(C, cin, 3))
    for i in range(1, 4):
        add("refer_enc.%d." % i, attention_block_spec(C, H))
    spec.append(("refer_enc.4.latents", (32, C)))
    for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
        add("refer_enc.4.cross_attention.%s." % n, conv(C, C, 1))
    add("refer_enc.4.enc.0.", conv(C, C, 3))
    for i in range(1, 5):
        add("refer_enc.4.enc.%d." % i, attention_block_spec(C, 8))
    add("integrating_conv.", conv(C, 2 * C, 1))
    for i in range(L):
        add("layers.%d." % i, dl())
    for i in range(L, L + 3):
        add("layers.%d." % i, res_block_spec(C, C))
    add("out.0.", [("weight", (C,)), ("bias", (C,))])
    add("out.2.", conv(cout, C, 3))
    return spec


# ---- model ---------------------------------------------------------------------------------------------------------------
def norm_groups(channels):
    """utils.py:118-133."""
    groups = 32
    if channels <= 16:
        groups = 8
    elif channels <= 64:
        groups = 16
    while channels % groups != 0:
        groups = int(groups / 2)
    return groups


def group_norm(x, sd, pfx):
    return F.group_norm(x.float(), norm_groups(x.shape[1]), sd[pfx + "weight"], sd[pfx + "bias"], 1e-5)


def conv1d(x, sd, pfx, padding=0):
    return F.conv1d(x, sd[pfx + "weight"], sd[pfx + "bias"], padding=padding)


def timestep_embedding(timesteps, dim, max_period=10000):
    """aa_model.py:32-51."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def relative_position_bucket(relative_position, num_buckets=32, max_distance=64):
    """xtransformers.py:155-174 with causal=False."""
    n = -relative_position
    num_buckets //= 2
    ret = (n < 0).long() * num_buckets
    n = torch.abs(n)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    val_if_large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                                * (num_buckets - max_exact)).long()
    val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, num_buckets - 1))
    return ret + torch.where(is_small, n, val_if_large)


def relative_position_bias(table, i, j, scale):
    """xtransformers.py:176-185: (1, H, i, j) additive logits; table (num_buckets, H)."""
    rel = torch.arange(j)[None, :] - torch.arange(i)[:, None]
    values = F.embedding(relative_position_bucket(rel, table.shape[0], 64), table)       # (i, j, H)
    return values.permute(2, 0, 1)[None] * scale


def attention_block(x, sd, pfx, heads):
    """utils.py:172-215 with QKVAttentionLegacy :136-169 and relative position embeddings."""
    b, c, t = x.shape
    qkv = conv1d(group_norm(x, sd, pfx + "norm."), sd, pfx + "qkv.")
    ch = c // heads
    q, k, v = qkv.reshape(b * heads, ch * 3, t).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = w.reshape(b, heads, t, t) + relative_position_bias(sd[pfx + "relative_pos_embeddings.relative_attention_bias.weight"],
                                                           t, t, ch ** 0.5)
    w = torch.softmax(w.reshape(b * heads, t, t).float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, t)
    return x + conv1d(a, sd, pfx + "proj_out.")


def res_block(x, emb, sd, pfx):
    """aa_model.py:70-131 (dims=1, efficient_config -> 1x1 in conv, kernel 3 out conv, use_scale_shift_norm=True)."""
    h = conv1d(F.silu(group_norm(x, sd, pfx + "in_layers.0.")), sd, pfx + "in_layers.2.")
    e = F.linear(F.silu(emb), sd[pfx + "emb_layers.1.weight"], sd[pfx + "emb_layers.1.bias"])[..., None]
    scale, shift = torch.chunk(e, 2, dim=1)
    h = group_norm(h, sd, pfx + "out_layers.0.") * (1 + scale) + shift
    h = conv1d(F.silu(h), sd, pfx + "out_layers.3.", padding=1)
    return x + h            # out_channels == channels on this path: skip_connection is the identity


def diffusion_layer(x, emb, sd, pfx, heads):
    """aa_model.py:134-148 (refer=None)."""
    return attention_block(res_block(x, emb, sd, pfx + "resblk."), sd, pfx + "attn.", heads)


def cross_attention(x, c, sd, pfx, heads):
    """vc_utils.MultiHeadAttention.forward/attention without window / mask (vc_utils.py:568-627)."""
    q, k, v = conv1d(x, sd, pfx + "conv_q."), conv1d(c, sd, pfx + "conv_k."), conv1d(c, sd, pfx + "conv_v.")
    b, d, tt = q.shape
    ts = k.shape[2]
    kc = d // heads
    qh = q.view(b, heads, kc, tt).transpose(2, 3)
    kh = k.view(b, heads, kc, ts).transpose(2, 3)
    vh = v.view(b, heads, kc, ts).transpose(2, 3)
    p = torch.softmax(torch.matmul(qh / math.sqrt(kc), kh.transpose(-2, -1)), dim=-1)
    o = torch.matmul(p, vh).transpose(2, 3).contiguous().view(b, d, tt)
    return conv1d(o, sd, pfx + "conv_o.")


def ref_encoder(x, sd, pfx, heads=8):
    """aa_model.py:150-177.  Note `latents[:, :self.latents.shape[1], :]` (:175) slices the CHANNEL axis with ref_dim, a no-op
    here (ref_dim == dim), so the mean (:176) runs over the 32 latent slots AND every reference frame."""
    lat = sd[pfx + "latents"]
    latents = lat.t()[None].expand(x.shape[0], -1, -1)
    latents = cross_attention(latents, x, sd, pfx + "cross_attention.", heads)
    h = conv1d(torch.cat((latents, x), -1), sd, pfx + "enc.0.", padding=1)
    for i in range(1, 5):
        h = attention_block(h, sd, pfx + "enc.%d." % i, heads)
    return torch.mean(h[:, :lat.shape[1], :], -1)


def aa_diffusion_forward(sd, cfg, x, timesteps, latent, refer, uncond=None, drop_layers=()):
    """AA_diffusion.forward (conditioning_free=False) (aa_model.py:238-287)."""
    C, H, L = cfg["model_channels"], cfg["num_heads"], cfg["num_layers"]
    # timestep_independent (:238-252)
    h = conv1d(latent, sd, "latent_conditioner.0.", padding=1)
    for i in range(1, 4):
        h = attention_block(h, sd, "latent_conditioner.%d." % i, H)
    r = conv1d(refer, sd, "refer_enc.0.", padding=1)
    for i in range(1, 4):
        r = attention_block(r, sd, "refer_enc.%d." % i, H)
    r = ref_encoder(r, sd, "refer_enc.4.")
    latent_emb = group_norm(h, sd, "code_norm.") + r.unsqueeze(-1)
    if uncond is not None:
        latent_emb = torch.where(uncond.view(-1, 1, 1), sd["unconditioned_embedding"].repeat(x.shape[0], 1, 1), latent_emb)
    latent_emb = F.interpolate(latent_emb, size=x.shape[-1], mode="nearest")
    # forward (:253-287)
    te = timestep_embedding(timesteps, C)
    te = F.linear(F.silu(F.linear(te, sd["time_embed.0.weight"], sd["time_embed.0.bias"])), sd["time_embed.2.weight"],
                  sd["time_embed.2.bias"])
    for i in range(3):
        latent_emb = diffusion_layer(latent_emb, te, sd, "conditioning_timestep_integrator.%d." % i, H)
    h = conv1d(x, sd, "inp_block.", padding=1)
    h = conv1d(torch.cat([h, latent_emb], dim=1), sd, "integrating_conv.")
    for i in range(L + 3):
        if i in drop_layers and i != 0 and i != L + 2:
            continue
        h = diffusion_layer(h, te, sd, "layers.%d." % i, H) if i < L else res_block(h, te, sd, "layers.%d." % i)
    return conv1d(F.silu(group_norm(h.float(), sd, "out.0.")), sd, "out.2.", padding=1)


# ---- Gaussian diffusion (linear betas, epsilon prediction, learned-range variance, mse loss) ------------------------------------
def diffusion_tables(num_steps=1000):
    """GaussianDiffusion.__init__ (diffusion.py:200-231) as SpacedDiffusion builds it when every step is kept
    (space_timesteps(n, [n]); :1181-1196): the linear betas (:83-98) are RE-DERIVED from their own cumulative product
    (`1 - alpha_cumprod / last_alpha_cumprod`), which moves them by a few float64 ulps.  All float64."""
    scale = 1000 / num_steps
    base = np.linspace(scale * 0.0001, scale * 0.02, num_steps, dtype=np.float64)
    base_ac = np.cumprod(1.0 - base, axis=0)
    betas, last = [], 1.0
    for a in base_ac:
        betas.append(1 - a / last)
        last = a
    betas = np.array(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {"betas": betas, "sqrt_alphas_cumprod": np.sqrt(ac), "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
            "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac), "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
            "posterior_variance": post_var,
            "posterior_log_variance_clipped": np.log(np.append(post_var[1], post_var[1:])),
            "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
            "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac), "log_betas": np.log(betas)}


def _ext(arr, t, ndim=3):
    r = torch.from_numpy(arr)[t].float()
    while r.dim() < ndim:
        r = r[..., None]
    return r


def q_sample(tab, x_start, t, noise):
    return _ext(tab["sqrt_alphas_cumprod"], t) * x_start + _ext(tab["sqrt_one_minus_alphas_cumprod"], t) * noise


def approx_standard_normal_cdf(x):
    return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * torch.pow(x, 3))))


def discretized_gaussian_log_likelihood(x, means, log_scales):
    centered = x - means
    inv_stdv = torch.exp(-log_scales)
    cdf_plus = approx_standard_normal_cdf(inv_stdv * (centered + 1.0 / 255.0))
    cdf_min = approx_standard_normal_cdf(inv_stdv * (centered - 1.0 / 255.0))
    log_cdf_plus = torch.log(cdf_plus.clamp(min=1e-12))
    log_one_minus_cdf_min = torch.log((1.0 - cdf_min).clamp(min=1e-12))
    cdf_delta = cdf_plus - cdf_min
    return torch.where(x < -0.999, log_cdf_plus,
                       torch.where(x > 0.999, log_one_minus_cdf_min, torch.log(cdf_delta.clamp(min=1e-12))))


def training_losses(tab, model_output, x_start, x_t, t, noise):
    """training_losses (mse + learned-range vb; diffusion.py:963-1010) given the model output (B, 2C, T).
    Returns dict(loss, mse, vb) of shape (B,).  The vb term sees the mean prediction detached (:980)."""
    C = x_start.shape[1]
    eps, var_values = torch.split(model_output, C, dim=1)
    eps_d = eps.detach()
    # p_mean_variance with clip_denoised=True (:316-392)
    min_log = _ext(tab["posterior_log_variance_clipped"], t)
    max_log = _ext(tab["log_betas"], t)
    frac = (var_values + 1) / 2
    log_var = frac * max_log + (1 - frac) * min_log
    pred_xstart = (_ext(tab["sqrt_recip_alphas_cumprod"], t) * x_t - _ext(tab["sqrt_recipm1_alphas_cumprod"], t) * eps_d).clamp(-1, 1)
    mean = _ext(tab["posterior_mean_coef1"], t) * pred_xstart + _ext(tab["posterior_mean_coef2"], t) * x_t
    true_mean = _ext(tab["posterior_mean_coef1"], t) * x_start + _ext(tab["posterior_mean_coef2"], t) * x_t
    true_log_var = _ext(tab["posterior_log_variance_clipped"], t).expand_as(x_t)
    kl = 0.5 * (-1.0 + log_var - true_log_var + torch.exp(true_log_var - log_var) + ((true_mean - mean) ** 2) * torch.exp(-log_var))
    kl = kl.mean(dim=(1, 2)) / np.log(2.0)
    nll = -discretized_gaussian_log_likelihood(x_start, mean, 0.5 * log_var).mean(dim=(1, 2)) / np.log(2.0)
    vb = torch.where(t == 0, nll, kl)
    mse = ((noise - eps) ** 2).mean(dim=(1, 2))
    return {"loss": mse + vb, "mse": mse, "vb": vb}


def warmup(step):
    """diffusion/train.py:69-73."""
    return float(step / 1000) if step < 1000 else 1
