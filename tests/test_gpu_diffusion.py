"""GPU parity of the diffusion mel-denoiser step (csrc/diffusion_ops.hip + conv / attention families through the C ABI,
ttts_amd/diffusion/) against torch fp32 references of single ops, the oracle (oracle/diffusion_ref.py) and the
reference-generated fixture tests/golden/diffusion.npz."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import diffusion_ref as DR

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "diffusion.npz")


@pytest.fixture(autouse=True)
def _exact_convs():
    """fp32-tolerance parity: run the exact convolution kernels (see tests/test_gpu_vqvae.py)."""
    from ttts_amd import ops as _ops
    _ops.set_conv_precision("exact")
    yield
    _ops.set_conv_precision("split_bf16")


def _dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def T(a):
    return torch.from_numpy(np.asarray(a))


def _load_det(module, gain=1.0, prefix=""):
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(DR.det_fill(prefix + k, p.shape, gain))


def _close(a, b, rtol, atol=0.0, msg=""):
    a = torch.as_tensor(a).detach().cpu().double(); b = torch.as_tensor(b).detach().cpu().double()
    err = (a - b).abs().max().item()
    assert err <= atol + rtol * b.abs().max().item(), "%s: max err %.3e vs ref max %.3e" % (msg, err, b.abs().max().item())


@pytest.mark.parametrize("B,C,Tn,G,ss,silu", [(2, 64, 37, 16, False, False), (3, 64, 50, 16, True, True), (2, 512, 100, 32, True, True),
                                              (2, 100, 33, 4, False, True), (1, 512, 400, 32, False, False),
                                              # (round 6: the register-resident kernels take T % 4 == 0 groups of <= 8192 elements /
                                              # <= 16 channels x 448 frames; the last case is past both limits: the three-pass fallback)
                                              (2, 64, 48, 8, True, True), (3, 128, 448, 8, True, False), (1, 1024, 400, 32, True, True)])
def test_groupnorm_vs_torch(B, C, Tn, G, ss, silu):
    from ttts_amd.diffusion.aa_model import _GroupNormFn
    g = torch.Generator().manual_seed(C + Tn)
    x = (torch.randn(B, C, Tn, generator=g) * 1.3 + 0.2).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True); beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    s = (0.3 * torch.randn(B, 2 * C, generator=g)).requires_grad_(True) if ss else None
    w = torch.randn(B, C, Tn, generator=g)
    y = F.group_norm(x, G, gamma, beta, 1e-5)
    if ss:
        y = y * (1 + s[:, :C, None]) + s[:, C:, None]
    if silu:
        y = F.silu(y)
    (y * w).sum().backward()
    dev = _dev()
    xd, gd, bd = x.detach().to(dev).requires_grad_(True), gamma.detach().to(dev).requires_grad_(True), beta.detach().to(dev).requires_grad_(True)
    sd = s.detach().to(dev).requires_grad_(True) if ss else None
    yd = _GroupNormFn.apply(xd, gd, bd, sd, G, silu)
    (yd * w.to(dev)).sum().backward()
    _close(yd, y, 1e-5, 1e-6, "y"); _close(xd.grad, x.grad, 2e-5, 1e-6, "dx")
    _close(gd.grad, gamma.grad, 2e-5, 1e-5, "dgamma"); _close(bd.grad, beta.grad, 2e-5, 1e-5, "dbeta")
    if ss:
        _close(sd.grad, s.grad, 2e-5, 1e-5, "dss")


def test_small_ops_vs_torch():
    from ttts_amd import ops
    from ttts_amd.diffusion.aa_model import _bucket_table
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    # timestep embedding
    t = torch.tensor([0, 1, 17, 500, 999])
    _close(ops.timestep_embedding(t.to(dev), 64), DR.timestep_embedding(t, 64), 0, 4e-6, "timestep_embedding")
    _close(ops.timestep_embedding(t.to(dev), 512), DR.timestep_embedding(t, 512), 0, 4e-6, "timestep_embedding 512")
    # nearest interpolation and its adjoint (integer and non-integer ratios)
    for tin, tout in ((12, 48), (100, 400), (7, 20), (30, 11)):
        x = torch.randn(2, 5, tin, generator=g).requires_grad_(True)
        y = F.interpolate(x, size=tout, mode="nearest")
        w = torch.randn(2, 5, tout, generator=g)
        (y * w).sum().backward()
        _close(ops.interp_nearest_fwd(x.detach().to(dev), tout), y, 0, 0, "interp %d->%d" % (tin, tout))
        _close(ops.interp_nearest_bwd(w.to(dev), tin), x.grad, 1e-6, 1e-6, "interp adjoint")
    # bucket table and bias
    for n in (5, 37, 130):
        rel = torch.arange(n)[None, :] - torch.arange(n)[:, None]
        want = DR.relative_position_bucket(rel, 32, 64)
        tab = _bucket_table(n, 32, 64, dev).cpu()
        off = (tab.numel() - 1) // 2
        assert torch.equal(tab[(rel + off).long()].long(), want)
    table = torch.randn(32, 4, generator=g)
    bias = ops.relpos_bias_fwd(table.to(dev), _bucket_table(37, 32, 64, dev), 4, 37, 37, 2.0)
    _close(bias, DR.relative_position_bias(table, 37, 37, 2.0)[0], 0, 1e-6, "relpos bias")
    dS = torch.randn(3, 4, 37, 37, generator=g)
    want = torch.zeros(32, 4)
    bk = DR.relative_position_bucket(torch.arange(37)[None, :] - torch.arange(37)[:, None], 32, 64)
    want.index_put_((bk.reshape(-1).repeat(4), torch.arange(4).repeat_interleave(37 * 37)), dS.sum(0).reshape(-1) * 2.0, accumulate=True)
    _close(ops.relpos_bias_bwd(dS.to(dev), _bucket_table(37, 32, 64, dev), 32, 2.0), want, 2e-5, 1e-5, "relpos dtable")
    # softmax with bias
    S = torch.randn(2, 4, 9, 21, generator=g); b = torch.randn(4, 9, 21, generator=g)
    _close(ops.softmax_bias_fwd(S.clone().to(dev), b.to(dev)), torch.softmax(S + b, -1), 1e-6, 1e-7, "softmax bias")
    # select rows
    use = torch.tensor([1, 0, 1], dtype=torch.uint8); a = torch.randn(3, 6, 10, generator=g); vec = torch.randn(6, generator=g)
    out = ops.select_rows_fwd(use.to(dev), a.to(dev), vec.to(dev))
    want = torch.where(use.bool().view(3, 1, 1), vec.view(1, 6, 1).expand(3, 6, 10), a)
    _close(out, want, 0, 0, "select")
    da, dvec = ops.select_rows_bwd(use.to(dev), a.to(dev))
    _close(da, a * (1 - use.float().view(3, 1, 1)), 0, 0, "select da"); _close(dvec, (a * use.float().view(3, 1, 1)).sum((0, 2)), 1e-6, 1e-6, "select dvec")


def test_diffusion_loss_vs_oracle():
    from ttts_amd import ops
    from ttts_amd.diffusion.gaussian import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    dev = _dev()
    d = SpacedDiffusion(space_timesteps(1000, [1000]), betas=get_named_beta_schedule("linear", 1000))
    tab = DR.diffusion_tables(1000)
    for k in ("sqrt_alphas_cumprod", "posterior_log_variance_clipped", "posterior_mean_coef2"):
        np.testing.assert_array_equal(getattr(d, k), tab[k])
    g = torch.Generator().manual_seed(9)
    B, C, Tn = 5, 20, 48
    x0 = (torch.randn(B, C, Tn, generator=g) * 0.5); x0[0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995]); x0[1, 0, :2] = torch.tensor([-1.0, 1.0])
    t = torch.tensor([0, 0, 1, 500, 999]); noise = torch.randn(B, C, Tn, generator=g)
    mo = (torch.randn(B, 2 * C, Tn, generator=g) * 0.8).requires_grad_(True)
    x_t = DR.q_sample(tab, x0, t, noise)
    _close(d.q_sample(x0.to(dev), t.to(dev), noise.to(dev)), x_t, 1e-6, 1e-7, "q_sample")
    terms = DR.training_losses(tab, mo, x0, x_t, t, noise)
    terms["loss"].mean().backward()
    tt, lm = ops.diffusion_loss_fwd(mo.detach().to(dev), x0.to(dev), x_t.to(dev), noise.to(dev), t.to(dev), d.table(dev))
    _close(tt[:, 0], terms["mse"], 1e-5, 0, "mse"); _close(tt[:, 1], terms["vb"], 2e-5, 1e-9, "vb"); _close(lm[0], terms["loss"].mean(), 2e-5, 0, "mean")
    dmo = ops.diffusion_loss_bwd(mo.detach().to(dev), x0.to(dev), x_t.to(dev), noise.to(dev), t.to(dev), d.table(dev))
    _close(dmo[:, :C], mo.grad[:, :C], 1e-5, 1e-9, "d eps"); _close(dmo[:, C:], mo.grad[:, C:], 5e-5, 1e-9, "d var")


def _attn_relpos_reference(qkv, table, bucket, H, scale):
    """QKVAttentionLegacy + RelativePositionBias in fp64 (ttts/utils/utils.py:136-169, xtransformers.py:176-185)."""
    B, W, Tn = qkv.shape
    ch = W // (3 * H)
    q, k, v = qkv.double().reshape(B * H, 3 * ch, Tn).split(ch, dim=1)
    sc = 1.0 / np.sqrt(np.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * sc, k * sc)
    off = (bucket.numel() - 1) // 2
    idx = (torch.arange(Tn)[None, :] - torch.arange(Tn)[:, None]) + off                  # d = key - query
    bias = table.double()[bucket.long()[idx]].permute(2, 0, 1) * scale                   # (H, Tq, Tk)
    w = (w.reshape(B, H, Tn, Tn) + bias[None]).reshape(B * H, Tn, Tn)
    w = torch.softmax(w, dim=-1)
    return torch.einsum("bts,bcs->bct", w, v).reshape(B, H * ch, Tn)


@pytest.mark.parametrize("B,H,Tn", [(2, 16, 400), (1, 16, 232), (3, 4, 100), (2, 3, 77), (1, 2, 33), (2, 16, 200), (1, 1, 448)])
@pytest.mark.parametrize("products", [3, 1])
def test_fused_relpos_attention_vs_fp64(B, H, Tn, products):
    """csrc/attn_relpos.hip (one forward, three backward launches; no (B, H, T, T) tensor) against the fp64 formula of the
    reference's attention: output, dqkv and the bias table's gradient.  Stated tolerances, relative to each tensor's range:
    split-bf16 operands (products = 3, the default path) 1e-4 (measured <= 4e-5) -- fp32-equivalent; plain bf16 operands (products = 1, the fp8 mode's
    autocast arithmetic) 2e-2.  dqkv is bit-reproducible, the bias gradient to fp32 summation noise."""
    from ttts_amd import ops
    from ttts_amd.diffusion.aa_model import _bucket_table
    g = torch.Generator().manual_seed(Tn + H)
    ch = 32
    qkv = torch.randn(B, 3 * H * ch, Tn, generator=g) * 1.2
    table = torch.randn(32, H, generator=g) * 0.5
    scale = float(ch) ** 0.5
    bucket = _bucket_table(Tn, 32, 64, torch.device("cpu"))
    q64 = qkv.double().requires_grad_(True); t64 = table.double().requires_grad_(True)
    ref = _attn_relpos_reference(q64, t64, bucket, H, scale)
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout.double())
    qd, td, bd, dd = qkv.to(_dev()), table.to(_dev()), bucket.to(_dev()), dout.to(_dev())
    out, lse = ops.attn_relpos_fwd(qd, td, bd, H, scale, products)
    dqkv, dtable = ops.attn_relpos_bwd(qd, td, bd, out, dd, lse, H, scale, products)
    dqkv2, dtable2 = ops.attn_relpos_bwd(qd, td, bd, out, dd, lse, H, scale, products)
    tol = 1e-4 if products == 3 else 2e-2
    _close(out, ref, tol, msg="out")
    _close(dqkv, q64.grad, tol, msg="dqkv")
    _close(dtable, t64.grad, tol, msg="dtable")
    # dqkv is bit-reproducible; the bias gradient's per-workgroup diagonal sums are LDS float atomics (order not fixed), the rest
    # of its reduction is ordered: run-to-run differences stay at fp32 summation noise
    assert torch.equal(dqkv, dqkv2)
    _close(dtable, dtable2, 2e-6, msg="dtable run to run")
    slot = torch.full_like(td, 2.0)                                   # accumulate into a gradient-arena slot
    ops.attn_relpos_bwd(qd, td, bd, out, dd, lse, H, scale, products, dtable=slot)
    _close(slot, dtable + 2.0, 2e-6, msg="accumulated dtable")
    none_q, none_t = ops.attn_relpos_bwd(qd, td, bd, out, dd, lse, H, scale, products, need_dtable=False)
    assert none_t is None and torch.equal(none_q, dqkv)


def test_attention_block_fused_equals_materialised_path(monkeypatch):
    """AttentionBlock through the fused kernels == through the batched-GEMM / softmax path it replaces (both fp32-equivalent):
    output and every gradient within 3e-5 of range."""
    from ttts_amd.diffusion import aa_model as M
    torch.manual_seed(3)
    blk = M.AttentionBlock(512, 16, relative_pos_embeddings=True).to(_dev())
    with torch.no_grad():
        blk.proj_out.weight.normal_(0, 0.05); blk.proj_out.bias.normal_(0, 0.05)
    x = torch.randn(2, 512, 232, device=_dev())
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(M, "_FUSED_ATTN", fused)
        xi = x.clone().requires_grad_(True)
        for prm in blk.parameters():
            prm.grad = None
        y = blk(xi)
        (y * torch.cos(y.detach())).sum().backward()
        res[fused] = [y.detach(), xi.grad] + [prm.grad.clone() for prm in blk.parameters()]
    for a, b in zip(res[True], res[False]):
        _close(a, b, 3e-5, msg="fused vs materialised")


def test_attention_block_and_res_block_match_fixture(gold):
    from ttts_amd.diffusion.aa_model import AttentionBlock, ResBlock
    dev = _dev()
    ab = AttentionBlock(64, 4, relative_pos_embeddings=True).to(dev); _load_det(ab)
    x = T(gold["ab_x"]).to(dev).requires_grad_(True)
    y = ab(x)
    _close(y, gold["ab_y"], 2e-5, 1e-6, "ab y")
    (y * torch.linspace(-1, 1, 37, device=dev)).sum().backward()
    _close(x.grad, gold["ab_dx"], 5e-5, 1e-6, "ab dx")
    _close(ab.relative_pos_embeddings.relative_attention_bias.weight.grad, gold["ab_dtable"], 1e-4, 1e-6, "ab dtable")
    _close(ab.qkv.weight.grad, gold["ab_dqkv_w"], 1e-4, 1e-6, "ab dqkv")
    rb = ResBlock(64, 64, 0, dims=1, use_scale_shift_norm=True).to(dev); _load_det(rb)
    x = T(gold["rb_x"]).to(dev).requires_grad_(True); e = T(gold["rb_emb"]).to(dev).requires_grad_(True)
    y = rb(x, e)
    _close(y, gold["rb_y"], 2e-5, 1e-6, "rb y")
    (y * torch.linspace(-1, 1, 29, device=dev)).sum().backward()
    _close(x.grad, gold["rb_dx"], 5e-5, 1e-6, "rb dx"); _close(e.grad, gold["rb_demb"], 1e-4, 1e-6, "rb demb")
    _close(rb.out_layers[0].weight.grad, gold["rb_dgamma_out"], 1e-4, 1e-6, "rb dgamma")


def _tiny(gold, train=True):
    from ttts_amd.diffusion import AA_diffusion
    cfg = json.loads(str(gold["cfg"]))
    m = AA_diffusion(**cfg).to(_dev())
    assert [k for k, _ in m.named_parameters()] == json.loads(str(gold["param_names"]))
    _load_det(m, 0.7)
    m.train(train)
    return cfg, m


def test_model_forward_loss_and_gradients_match_fixture(gold):
    from ttts_amd.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    dev = _dev()
    cfg, m = _tiny(gold)
    d = SpacedDiffusion(space_timesteps(1000, [1000]), betas=get_named_beta_schedule("linear", 1000))
    D = lambda k: T(gold[k]).to(dev)
    out = d.training_losses(m, D("x_start"), D("t"), model_kwargs={"latent": D("latent"), "refer": D("refer")}, noise=D("noise"))
    for k in ("loss", "mse", "vb"):
        _close(out[k], gold[k], 5e-5, 1e-8, k)
    with torch.no_grad():
        mo = m(D("x_t"), D("t"), latent=D("latent"), refer=D("refer"))
    _close(mo, gold["model_out"], 1e-4, 1e-5, "model_out")
    out["loss_mean"].backward()
    names = json.loads(str(gold["param_names"]))
    ps = dict(m.named_parameters())
    for i, k in enumerate(names):
        g = ps[k].grad
        ref = gold["grad_abs_sum"][i]
        got = 0.0 if g is None else float(g.abs().sum())
        assert abs(got - ref) <= 3e-3 * ref + 1e-6, (k, got, ref)
    for k in [n[5:] for n in gold.files if n.startswith("grad:")]:
        _close(ps[k].grad, gold["grad:" + k], 5e-4, 1e-7, k)


def test_forced_random_branches_match_fixture(gold):
    dev = _dev()
    cfg, m = _tiny(gold, train=False)
    D = lambda k: T(gold[k]).to(dev)
    with torch.no_grad():
        a = m(D("x_t"), D("t"), latent=D("latent"), refer=D("refer"))
        b = m(D("x_t"), D("t"), latent=D("latent"), refer=D("refer"), uncond=torch.ones(3, dtype=torch.bool, device=dev))
        c = m(D("x_t"), D("t"), latent=D("latent"), refer=D("refer"), drop_layers={1, 2})
    _close(a, gold["model_out_eval"], 1e-4, 1e-5, "eval"); _close(b, gold["model_out_uncond"], 1e-4, 1e-5, "uncond")
    sd = {k: v.detach().cpu() for k, v in m.named_parameters()}
    want = DR.aa_diffusion_forward(sd, cfg, T(gold["x_t"]), T(gold["t"]), T(gold["latent"]), T(gold["refer"]), drop_layers={1, 2})
    _close(c, want, 1e-4, 1e-5, "layer drop")


def test_trainer_three_steps_match_fixture(gold):
    from ttts_amd.diffusion.train import DiffusionTrainer
    dev = _dev()
    cfg = {"train": {"lr": 1e-4, "timesteps": 1000}, "aa_diffusion": json.loads(str(gold["cfg"]))}
    tr = DiffusionTrainer(cfg, device=dev)
    _load_det(tr.diffusion, 0.7)
    before = {k: p.detach().clone() for k, p in tr.diffusion.named_parameters()}
    tr.step = 1                                                    # the fixture starts after the lr-0 step of LambdaLR
    D = lambda k: T(gold[k]).to(dev)
    norms, losses = [], []
    for _ in range(3):
        out = tr.train_step(D("x_start"), D("refer"), D("latent"), t=D("t"), noise=D("noise"), normalized=True)
        norms.append(float(out["grad_norm"])); losses.append(float(out["loss"]))
    np.testing.assert_allclose(losses, gold["step_losses"], rtol=1e-4)
    np.testing.assert_allclose(norms, gold["step_norms"], rtol=2e-3)
    delta = np.array([float((p.detach() - before[k]).abs().sum()) for k, p in tr.diffusion.named_parameters()])
    ref = gold["step_delta_abs"]
    big = ref > 1e-3 * ref.max()
    assert np.abs(delta[big] - ref[big]).max() <= 3e-2 * ref.max()
    assert np.median(np.abs(delta[big] / ref[big] - 1)) < 1e-2


def test_graphed_diffusion_step_follows_the_eager_step():
    """DiffusionTrainer.train_step_graphed (one hipGraph per layer-drop pattern, shared pool, device-side warm-up factor) against
    the launch-by-launch step: same seeds -> the same layer-drop draws, timesteps and noise -> losses and gradient norms of six
    steps equal to 2e-5 relative (the learning rate's warm-up factor is rounded on the device instead of the host), parameters after
    the six steps to 1e-6 of their range; at least two different recordings were replayed."""
    import random
    from ttts_amd.diffusion.train import DiffusionTrainer
    cfg = {"train": {"lr": 1e-4, "timesteps": 1000},
           "aa_diffusion": dict(in_channels=100, out_channels=200, model_channels=512, num_heads=16, num_layers=4, in_latent_channels=512,
                                dropout=0, layer_drop=0.35, unconditioned_percentage=0.0)}
    g = torch.Generator().manual_seed(1)
    mel = (torch.randn(2, 100, 120, generator=g) * 2 - 4).to(_dev()); ref = (torch.randn(2, 100, 80, generator=g) * 2 - 4).to(_dev())
    lat = torch.randn(2, 512, 30, generator=g).to(_dev())
    res = {}
    for graphed in (False, True):
        random.seed(11); torch.manual_seed(11)
        tr = DiffusionTrainer(cfg, device=_dev(), seed=3)
        with torch.no_grad():
            for k, p in tr.diffusion.named_parameters():
                if k.endswith("proj_out.weight"):
                    p.normal_(0, 0.02)
        vals = []
        for i in range(8):
            fn = tr.train_step_graphed if (graphed and i >= 2) else tr.train_step
            out = fn(mel, ref, lat)
            vals.append((float(out["loss"]), float(out["grad_norm"])))
        res[graphed] = (vals, tr.optimizer.flat_p.clone(), tr)
    st = res[True][2]._gstate
    assert not st["failed"] and len(st["graphs"]) >= 6, (st["failed"], list(st["graphs"]))   # () and the five single drops
    np.testing.assert_allclose(np.array(res[True][0]), np.array(res[False][0]), rtol=2e-5)
    _close(res[True][1], res[False][1], 1e-6, msg="parameters after 8 steps")
    assert res[True][2].step == res[False][2].step == 8 and float(res[True][2].optimizer.opt_state[0]) == 8.0
