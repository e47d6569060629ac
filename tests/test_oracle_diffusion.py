"""CPU: the diffusion restatement (oracle/diffusion_ref.py) against the reference-generated fixture tests/golden/diffusion.npz
(tools/make_goldens.py diffusion; reference ttts/diffusion/aa_model.py, ttts/utils/{utils,diffusion,xtransformers}.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_ref as D

GOLD = os.path.join(os.path.dirname(__file__), "golden", "diffusion.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def tiny(gold):
    cfg = json.loads(str(gold["cfg"]))
    names = json.loads(str(gold["param_names"]))
    shapes = _tiny_shapes(cfg)
    assert list(shapes) == names
    return cfg, {k: D.det_fill(k, s, 0.7) for k, s in shapes.items()}


def _tiny_shapes(cfg):
    """Parameter names / shapes of AA_diffusion(**cfg) in registration order (pinned against the fixture's name list)."""
    return dict(D.param_spec(cfg))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_full_config_surface(gold):
    full = json.loads(str(gold["full_cfg"]))
    surf = json.loads(str(gold["full_surface"]))
    spec = D.param_spec(full)
    assert [[k, list(s)] for k, s in spec] == surf                  # 283 tensors: every state-dict entry is a parameter
    assert sum(int(np.prod(s)) for _, s in spec) == 43222728


def test_attention_block_and_res_block(gold):
    sd = {"a." + k: D.det_fill(k, s) for k, s in D.attention_block_spec(64, 4)}
    x = T(gold["ab_x"]).requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = D.attention_block(x, leaves, "a.", 4)
    np.testing.assert_allclose(y.detach().numpy(), gold["ab_y"], atol=2e-5)
    (y * torch.linspace(-1, 1, 37)).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), gold["ab_dx"], atol=2e-5)
    np.testing.assert_allclose(leaves["a.relative_pos_embeddings.relative_attention_bias.weight"].grad.numpy(), gold["ab_dtable"], atol=2e-4)
    np.testing.assert_allclose(leaves["a.qkv.weight"].grad.numpy(), gold["ab_dqkv_w"], atol=2e-4)
    sd = {"r." + k: D.det_fill(k, s) for k, s in D.res_block_spec(64, 64)}
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = T(gold["rb_x"]).requires_grad_(True); e = T(gold["rb_emb"]).requires_grad_(True)
    y = D.res_block(x, e, leaves, "r.")
    np.testing.assert_allclose(y.detach().numpy(), gold["rb_y"], atol=2e-5)
    (y * torch.linspace(-1, 1, 29)).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), gold["rb_dx"], atol=2e-5)
    np.testing.assert_allclose(e.grad.numpy(), gold["rb_demb"], atol=2e-4)
    np.testing.assert_allclose(leaves["r.out_layers.0.weight"].grad.numpy(), gold["rb_dgamma_out"], atol=2e-4)


def test_tables_and_q_sample(gold):
    tab = D.diffusion_tables(1000)
    for k in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_log_variance_clipped", "posterior_mean_coef1",
              "posterior_mean_coef2", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        np.testing.assert_array_equal(tab[k], gold["tab:" + k])
    x_t = D.q_sample(tab, T(gold["x_start"]), T(gold["t"]), T(gold["noise"]))
    np.testing.assert_allclose(x_t.numpy(), gold["x_t"], atol=1e-6)


def test_model_forward_loss_and_gradients(gold, tiny):
    cfg, sd = tiny
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    tab = D.diffusion_tables(1000)
    x0, t, noise = T(gold["x_start"]), T(gold["t"]), T(gold["noise"])
    x_t = D.q_sample(tab, x0, t, noise)
    out = D.aa_diffusion_forward(leaves, cfg, x_t, t, T(gold["latent"]), T(gold["refer"]))
    np.testing.assert_allclose(out.detach().numpy(), gold["model_out"], atol=5e-5)
    terms = D.training_losses(tab, out, x0, x_t, t, noise)
    for k in ("loss", "mse", "vb"):
        np.testing.assert_allclose(terms[k].detach().numpy(), gold[k], rtol=2e-5, atol=1e-8)
    terms["loss"].mean().backward()
    names = json.loads(str(gold["param_names"]))
    for i, k in enumerate(names):
        g = leaves[k].grad
        if g is None:       # the reference adds `p.mean() * 0` for unused parameters (aa_model.py:281-285): a zero gradient
            assert gold["grad_abs_sum"][i] == 0.0, k
            continue
        assert abs(float(g.abs().sum()) - gold["grad_abs_sum"][i]) <= 2e-3 * gold["grad_abs_sum"][i] + 1e-6, k
    for k in [n[5:] for n in gold.files if n.startswith("grad:")]:
        ref = gold["grad:" + k]
        assert np.abs(leaves[k].grad.numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-7, k


def test_forced_random_branches(gold, tiny):
    cfg, sd = tiny
    tab = D.diffusion_tables(1000)
    x_t, t = T(gold["x_t"]), T(gold["t"])
    with torch.no_grad():
        a = D.aa_diffusion_forward(sd, cfg, x_t, t, T(gold["latent"]), T(gold["refer"]))
        b = D.aa_diffusion_forward(sd, cfg, x_t, t, T(gold["latent"]), T(gold["refer"]), uncond=torch.ones(3, dtype=torch.bool))
    np.testing.assert_allclose(a.numpy(), gold["model_out_eval"], atol=5e-5)
    np.testing.assert_allclose(b.numpy(), gold["model_out_uncond"], atol=5e-5)
    assert np.abs(gold["model_out_uncond"] - gold["model_out_eval"]).max() > 1e-3


def test_product_module_surface_matches_reference(gold):
    """ttts_amd.diffusion.AA_diffusion at the shipped config: state-dict keys, shapes and order of the reference (283 tensors)."""
    from ttts_amd.diffusion import AA_diffusion
    full = json.loads(str(gold["full_cfg"]))
    m = AA_diffusion(**full)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == json.loads(str(gold["full_surface"]))
    assert [k for k, _ in m.named_parameters()] == [k for k, _ in D.param_spec(full)]


def test_product_host_schedule_logic(gold):
    """Host side of ttts_amd.diffusion.gaussian (runs without a GPU): timestep spacing, beta schedules, the float64 coefficient
    tables SpacedDiffusion hands to the kernels -- against the reference-generated fixture and the oracle's tables."""
    from ttts_amd.diffusion.gaussian import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    assert sorted(space_timesteps(1000, [50])) == gold["space_50"].tolist()
    assert sorted(space_timesteps(1000, "ddim25")) == gold["space_ddim25"].tolist()
    assert sorted(space_timesteps(300, [10, 15, 20])) == gold["space_10_15_20"].tolist()
    np.testing.assert_allclose(get_named_beta_schedule("cosine", 100), gold["betas_cosine_100"], rtol=1e-12)
    d = SpacedDiffusion(space_timesteps(1000, [1000]), betas=get_named_beta_schedule("linear", 1000))
    tab = D.diffusion_tables(1000)
    cols = ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
            "posterior_mean_coef1", "posterior_mean_coef2", "posterior_log_variance_clipped", "log_betas")
    for i, k in enumerate(cols):
        np.testing.assert_array_equal(d._table_host[:, i], tab[k].astype(np.float32))
        if ("tab:" + k) in gold.files:
            np.testing.assert_array_equal(getattr(d, k), gold["tab:" + k])
    d50 = SpacedDiffusion(space_timesteps(1000, [50]), betas=get_named_beta_schedule("linear", 1000))
    np.testing.assert_array_equal(d50.betas, gold["spaced50_betas"])
    assert d50.timestep_map == gold["spaced50_map"].tolist() and d50.num_timesteps == 50
    np.testing.assert_array_equal(d50.posterior_log_variance_clipped, gold["spaced50_post_logvar"])
    with pytest.raises(NotImplementedError):
        SpacedDiffusion(space_timesteps(10, [10]), betas=np.full(10, 0.01), loss_type="kl")
