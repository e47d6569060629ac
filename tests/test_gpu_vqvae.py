"""GPU parity tests of the VQ-VAE-GAN conv family and modules (through the C ABI) against the CPU oracle / torch fp32
CPU reference of the same op and the reference-generated fixtures."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


def _close(a, b, rtol, atol, msg=""):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, "%s: max err %.3e vs ref max %.3e" % (msg, err, ref)


CONV_CASES = [  # Cin, Cout, K, stride, pad, dil, L, in_slope, resid, tanh
    (16, 32, 16, 10, 3, 1, 1000, 0.1, False, False),
    (32, 32, 3, 1, 3, 3, 300, 0.1, True, False),
    (32, 32, 7, 1, 15, 5, 300, 0.1, True, False),
    (48, 40, 11, 1, 5, 1, 257, 1.0, False, False),
    (5, 7, 5, 2, 2, 1, 131, 0.01, False, True),
    (4, 8, 41, 4, 20, 1, 500, 0.1, False, False),
    (1, 16, 7, 1, 3, 1, 700, 1.0, False, False),
    (16, 1, 7, 1, 3, 1, 700, 0.01, False, True),
    (192, 192, 2, 2, 0, 1, 64, 1.0, False, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d_family_vs_torch(case):
    from ttts_amd.vqvae.modules import _Conv1dFn
    cin, cout, k, s, pad, dil, L, slope, use_res, use_tanh = case
    g = torch.Generator().manual_seed(cin * 131 + k)
    B = 2
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    bb = torch.randn(B, cout, generator=g) * 0.1
    lout = (L + 2 * pad - dil * (k - 1) - 1) // s + 1
    res = torch.randn(B, cout, lout, generator=g) if use_res else None
    ct = torch.randn(B, cout, lout, generator=g)

    def ref(x, w, b, bb, res):
        y = F.conv1d(F.leaky_relu(x, slope), w, b, stride=s, padding=pad, dilation=dil) + bb[:, :, None]
        if res is not None:
            y = y + res
        return torch.tanh(y) if use_tanh else y

    leaves = [t.clone().requires_grad_(True) for t in (x, w, b, bb)] + ([res.clone().requires_grad_(True)] if use_res else [None])
    yr = ref(*leaves)
    (yr * ct).sum().backward()
    dl = [t.detach().to(_dev()).requires_grad_(True) if t is not None else None for t in (x, w, b, bb, res)]
    y = _Conv1dFn.apply(dl[0], dl[1], dl[2], dl[4], dl[3], s, pad, dil, float(slope), "tanh" if use_tanh else None)
    (y * ct.to(_dev())).sum().backward()
    _close(y, yr, 2e-5, 1e-6, "y")
    for name, a, r in zip(("dx", "dw", "db", "dbb", "dres"), dl, leaves):
        if a is not None:
            _close(a.grad, r.grad, 5e-5, 1e-6, name)


CONVT_CASES = [  # Cin, Cout, K, stride, pad, L, in_slope
    (32, 16, 16, 8, 4, 40, 0.1),
    (16, 8, 2, 2, 0, 333, 0.1),
    (24, 12, 8, 2, 3, 100, 0.1),
    (64, 32, 16, 10, 3, 32, 1.0),
    (7, 5, 4, 4, 0, 70, 0.1),
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose1d_vs_torch(case):
    from ttts_amd.vqvae.modules import _ConvTranspose1dFn
    cin, cout, k, s, pad, L, slope = case
    g = torch.Generator().manual_seed(cin * 17 + k)
    B = 3
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cin, cout, k, generator=g) / (cin * k / s) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (x, w, b)]
    yr = F.conv_transpose1d(F.leaky_relu(leaves[0], slope), leaves[1], leaves[2], stride=s, padding=pad)
    ct = torch.randn(yr.shape, generator=g)
    (yr * ct).sum().backward()
    dl = [t.detach().to(_dev()).requires_grad_(True) for t in (x, w, b)]
    y = _ConvTranspose1dFn.apply(dl[0], dl[1], dl[2], s, pad, float(slope))
    (y * ct.to(_dev())).sum().backward()
    _close(y, yr, 2e-5, 1e-6, "y")
    for name, a, r in zip(("dx", "dw", "db"), dl, leaves):
        _close(a.grad, r.grad, 5e-5, 1e-6, name)


def test_weight_norm_and_add_scale():
    from ttts_amd import ops
    from ttts_amd.vqvae.modules import _WeightNormFn, add_scale
    g = torch.Generator().manual_seed(3)
    v = torch.randn(37, 16, 7, generator=g); gg = torch.rand(37, 1, 1, generator=g) + 0.5
    ct = torch.randn(37, 16, 7, generator=g)
    vr, gr = v.clone().requires_grad_(True), gg.clone().requires_grad_(True)
    wr = gr * vr / vr.flatten(1).norm(dim=1).view(-1, 1, 1)
    (wr * ct).sum().backward()
    vd, gd = v.to(_dev()).requires_grad_(True), gg.to(_dev()).requires_grad_(True)
    w = _WeightNormFn.apply(vd, gd)
    (w * ct.to(_dev())).sum().backward()
    _close(w, wr, 1e-6, 1e-7, "w"); _close(vd.grad, vr.grad, 1e-5, 1e-7, "dv"); _close(gd.grad, gr.grad, 1e-5, 1e-7, "dg")
    xs = [torch.randn(3, 5, 1003, generator=g) for _ in range(3)]
    xd = [t.to(_dev()).requires_grad_(True) for t in xs]
    y = add_scale(xd, 1.0 / 3)
    assert torch.equal(y.cpu(), ((xs[0] + xs[1]) + xs[2]) * (1.0 / 3))
    y.sum().backward()
    assert torch.allclose(xd[1].grad.cpu(), torch.full((3, 5, 1003), 1.0 / 3))


def test_generator_matches_reference_fixture(golden_dir):
    """tests/golden/vqvae_generator.npz was produced by the imported reference Generator (tools/make_goldens.py G5)."""
    from ttts_amd.vqvae.modules import Generator
    g = np.load(os.path.join(golden_dir, "vqvae_generator.npz"))
    cfg = json.loads(str(g["cfg_json"]))
    m = Generator(**cfg)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:")})
    m = m.to(_dev())
    x = torch.from_numpy(g["x"]).to(_dev()).requires_grad_(True)
    gg = torch.from_numpy(g["g"]).to(_dev()).requires_grad_(True)
    y = m(x, gg)
    _close(y, torch.from_numpy(g["y"]), 2e-5, 2e-6, "y")
    (y * torch.from_numpy(g["ct"]).to(_dev())).sum().backward()
    _close(x.grad, torch.from_numpy(g["dx"]), 1e-4, 1e-6, "dx")
    _close(gg.grad, torch.from_numpy(g["dg"]), 1e-4, 1e-6, "dg")
    params = dict(m.named_parameters())
    assert set(params) == {k[5:] for k in g.files if k.startswith("grad:")}
    for k in g.files:
        if k.startswith("grad:"):
            _close(params[k[5:]].grad, torch.from_numpy(g[k]), 2e-4, 1e-6, k)


def test_resblock1_vs_oracle_at_path_shape():
    """ResBlock1(64, 11, (1,3,5)) at L=2048 (enc_q stage 2 of Appendix A) against oracle/vqvae_ref.py on the CPU."""
    from oracle import vqvae_ref
    from ttts_amd.vqvae.modules import ResBlock1
    torch.manual_seed(5)
    m = ResBlock1(64, 11, (1, 3, 5))
    sd = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x = torch.randn(2, 64, 2048)
    xr = x.clone().requires_grad_(True)
    yr = vqvae_ref.resblock1(xr, sd, "", 11, (1, 3, 5))
    yr.square().mean().backward()
    m = m.to(_dev())
    xd = x.to(_dev()).requires_grad_(True)
    y = m(xd)
    y.square().mean().backward()
    _close(y, yr, 2e-5, 1e-6, "y")
    _close(xd.grad, xr.grad, 1e-4, 1e-8, "dx")
    for k, p in m.named_parameters():
        _close(p.grad, sd[k].grad, 2e-4, 1e-8, k)


@pytest.mark.parametrize("case", [(16, 64, 41, 4, 20, 4, 600), (64, 256, 41, 4, 20, 16, 300), (512, 512, 41, 4, 20, 128, 90),
                                  (1, 16, 15, 1, 7, 1, 500), (96, 64, 5, 3, 2, 1, 200)])
def test_grouped_conv_with_lrelu_output_vs_torch(case):
    """DiscriminatorS / DiscriminatorP layers: (grouped) conv followed by leaky-relu(0.1) on the OUTPUT."""
    from ttts_amd.vqvae.modules import _Conv1dFn
    cin, cout, k, s, pad, G, L = case
    g = torch.Generator().manual_seed(cin + 7 * G)
    B = 2
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin // G, k, generator=g) / (cin // G * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (x, w, b)]
    yr = F.leaky_relu(F.conv1d(leaves[0], leaves[1], leaves[2], stride=s, padding=pad, groups=G), 0.1)
    ct = torch.randn(yr.shape, generator=g)
    (yr * ct).sum().backward()
    dl = [t.detach().to(_dev()).requires_grad_(True) for t in (x, w, b)]
    y = _Conv1dFn.apply(dl[0], dl[1], dl[2], None, None, s, pad, 1, 1.0, "lrelu", G, 0.1)
    (y * ct.to(_dev())).sum().backward()
    _close(y, yr, 2e-5, 1e-6, "y")
    for name, a, r in zip(("dx", "dw", "db"), dl, leaves):
        _close(a.grad, r.grad, 5e-5, 1e-6, name)


def test_discriminators_and_losses_match_reference_fixture(golden_dir):
    """MultiPeriodDiscriminator + losses.py through the HIP kernels vs tests/golden/vqvae_disc.npz (reference-generated;
    weights from oracle.vqvae_ref.det_fill)."""
    from oracle import vqvae_ref
    from ttts_amd.vqvae import losses as L
    from ttts_amd.vqvae.vq2 import MultiPeriodDiscriminator
    g = np.load(os.path.join(golden_dir, "vqvae_disc.npz"))
    mpd = MultiPeriodDiscriminator()
    mpd.load_state_dict({k: vqvae_ref.det_fill(k, v.shape) for k, v in mpd.state_dict().items()})
    mpd = mpd.to(_dev())
    y = torch.from_numpy(g["y"]).to(_dev()); y_hat = torch.from_numpy(g["y_hat"]).to(_dev()).requires_grad_(True)
    dr, dg, _, _ = mpd(y, y_hat.detach())
    loss_d, r_l, g_l = L.discriminator_loss(dr, dg)
    np.testing.assert_allclose(loss_d.item(), g["loss_disc"], rtol=2e-5)
    np.testing.assert_allclose([float(v) for v in r_l], g["r_losses"], rtol=2e-5)
    np.testing.assert_allclose([float(v) for v in g_l], g["g_losses"], rtol=2e-5)
    for i in range(6):
        _close(dr[i], torch.from_numpy(g[f"logit_r{i}"]), 5e-5, 1e-6, "logit_r%d" % i)
        _close(dg[i], torch.from_numpy(g[f"logit_g{i}"]), 5e-5, 1e-6, "logit_g%d" % i)
    loss_d.backward()
    np.testing.assert_allclose([p.grad.abs().sum().item() for _, p in mpd.named_parameters()], g["d_grad_abs_sum"], rtol=5e-4)
    np.testing.assert_allclose([p.grad.sum().item() for _, p in mpd.named_parameters()], g["d_grad_sum"], rtol=5e-3,
                               atol=1e-4 * float(np.abs(g["d_grad_abs_sum"]).max()))
    mpd.zero_grad()
    dr, dg, fr, fg = mpd(y, y_hat)
    shapes = json.loads(str(g["fmap_shapes"]))
    assert [[list(f.shape) for f in fl] for fl in fg] == shapes
    got = np.array([[f.abs().mean().item() for f in fl] + [0.0] * (7 - len(fl)) for fl in fg])
    np.testing.assert_allclose(got, g["fmap_abs_mean"], rtol=5e-5)
    lfm = L.feature_loss(fr, fg)
    lgen, gen_losses = L.generator_loss(dg)
    assert len(gen_losses) == 6
    np.testing.assert_allclose([lfm.item(), lgen.item()], [g["loss_fm"], g["loss_gen"]], rtol=2e-5)
    (lfm + lgen).backward()
    _close(y_hat.grad, torch.from_numpy(g["dy_hat"]), 1e-3, 1e-7, "dy_hat")
    zs = [torch.from_numpy(a).to(_dev()).requires_grad_(True) for a in g["kl_in"]]
    kl = L.kl_loss(*zs, torch.from_numpy(g["kl_mask"]).to(_dev()))
    np.testing.assert_allclose(kl.item(), g["kl"], rtol=2e-6)
    (kl * 3.0).backward()
    for z, want in zip(zs, g["kl_grads"]):
        _close(z.grad, torch.from_numpy(want) * 3.0, 1e-5, 1e-7, "kl grad")


def test_l1_loss_and_layout_agnostic_feature_pairs():
    from ttts_amd.vqvae import losses as L
    g = torch.Generator().manual_seed(9)
    a = torch.randn(2, 5, 33, 3, generator=g); b = torch.randn(2, 5, 33, 3, generator=g)
    # (B, C, H, W) views of (B, W, C, H) storage, as the period discriminators hand them out
    av = a.permute(0, 3, 1, 2).contiguous().to(_dev()).permute(0, 2, 3, 1)
    bv = b.permute(0, 3, 1, 2).contiguous().to(_dev()).permute(0, 2, 3, 1).requires_grad_(True)
    br = b.clone().requires_grad_(True)
    want = torch.mean(torch.abs(a - br)); want.backward()
    got = L.l1_loss(av, bv); got.backward()
    np.testing.assert_allclose(got.item(), want.item(), rtol=1e-6)
    _close(bv.grad, br.grad, 1e-6, 0, "dl1")
