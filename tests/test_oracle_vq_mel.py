"""Oracle VQ / mel restatements against the fixtures generated from the imported reference."""
import os

import numpy as np
import torch

from oracle import mel_ref, vq_ref

K, D = 1024, 192


def _sample(t, n):
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)].numpy()


def vq_case(g, name):
    seed, N, s = g[name + ":seed_N_scale"]
    rng = np.random.default_rng(int(seed))
    x = torch.from_numpy(rng.standard_normal((int(N), D), dtype=np.float32) * np.float32(s))
    e = torch.from_numpy(rng.standard_normal((K, D), dtype=np.float32))
    return x, e, g[name + ":idx"]


def test_vq_indices_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "vq.npz"))
    for name in ("n1024_s1", "n1024_s01", "n1024_s10", "n4096_s1"):
        x, e, idx = vq_case(g, name)
        got = vq_ref.quantize(x, e).numpy()
        assert np.array_equal(got, idx), name
        assert int(vq_ref.near_tie_audit(x, e, torch.from_numpy(idx)).sum()) == 0, name


def test_vq_ties_lowest_index(golden_dir):
    g = np.load(os.path.join(golden_dir, "vq.npz"))
    rng = np.random.default_rng(int(g["ties:seed"]))
    e = rng.standard_normal((K, D), dtype=np.float32)
    e[512:] = e[:512]
    x = e[rng.integers(0, 512, 256)] + rng.standard_normal((256, D), dtype=np.float32) * np.float32(0.01)
    got = vq_ref.quantize(torch.from_numpy(x), torch.from_numpy(e)).numpy()
    assert np.array_equal(got, g["ties:idx"]) and got.max() < 512


def test_vq_train_forward_and_ema(golden_dir):
    g = np.load(os.path.join(golden_dir, "vq.npz"))
    rng = np.random.default_rng(int(g["train:seed"]))
    e = torch.from_numpy(rng.standard_normal((K, D), dtype=np.float32))
    x = torch.from_numpy(rng.standard_normal((4, D, 128), dtype=np.float32)).requires_grad_(True)
    buf = {"embed": e.clone(), "embed_avg": e * 4.0, "cluster_size": torch.full((K,), 4.0)}
    q, codes, commit, ql = vq_ref.rvq_forward(x, buf, training=True)
    (q.sum() * 0.5 + commit).backward()
    assert np.array_equal(codes.numpy(), g["train:codes"])
    np.testing.assert_allclose(q.detach().numpy(), g["train:quantized"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(commit.item(), g["train:commit"], rtol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g["train:dx"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(buf["cluster_size"].numpy(), g["train:cluster_size"], rtol=1e-6)
    np.testing.assert_allclose(_sample(buf["embed_avg"], 8192), g["train:embed_avg_sample"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(_sample(buf["embed"], 8192), g["train:embed_sample"], rtol=1e-5, atol=1e-6)


def test_mel_set_a(golden_dir):
    g = np.load(os.path.join(golden_dir, "mel.npz"))
    y = torch.from_numpy(g["A:wav"])
    spec = mel_ref.spectrogram(y, 2048, 640, 2048)
    np.testing.assert_allclose(spec.numpy(), g["A:spec"], rtol=1e-5, atol=1e-6)
    mel = mel_ref.spec_to_mel(spec, 2048, 128, 32000, 0, None)
    np.testing.assert_allclose(mel.numpy(), g["A:mel"], rtol=1e-5, atol=1e-5)
    yseg = torch.from_numpy(g["A:wav"][:, :20480].copy()).requires_grad_(True)
    m = mel_ref.mel_spectrogram(yseg, 2048, 128, 32000, 640, 2048, 0, None)
    loss = torch.nn.functional.l1_loss(m, torch.from_numpy(g["A:seg_target"])) * 45
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["A:seg_loss"], rtol=1e-6)
    np.testing.assert_allclose(yseg.grad.numpy(), g["A:seg_grad"], rtol=1e-4, atol=1e-6)


def test_mel_set_b(golden_dir):
    g = np.load(os.path.join(golden_dir, "mel.npz"))
    y = torch.from_numpy(g["B:wav"])
    np.testing.assert_allclose(mel_ref.spectrogram(y, 1024, 256, 1024).numpy(), g["B:spec"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mel_ref.mel_spectrogram(y, 1024, 80, 22050, 256, 1024, 0, 8000).numpy(), g["B:mel"],
                               rtol=1e-5, atol=1e-5)


def test_generator_restatement(golden_dir):
    """oracle/vqvae_ref.py (ResBlock1, Generator, both weight-norm styles) vs the reference-generated fixture."""
    import json
    from oracle import vqvae_ref
    g = np.load(os.path.join(golden_dir, "vqvae_generator.npz"))
    cfg = json.loads(str(g["cfg_json"]))
    sd = {k[3:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("sd:")}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    gg = torch.from_numpy(g["g"]).requires_grad_(True)
    y = vqvae_ref.generator_forward(sd, cfg, x, gg)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-5, atol=1e-6)
    (y * torch.from_numpy(g["ct"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gg.grad.numpy(), g["dg"], rtol=1e-4, atol=1e-6)
    for k in g.files:
        if k.startswith("grad:"):
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)


def _mpd_state_dict():
    """Reference key/shape list (surface.json, G0) filled with oracle.vqvae_ref.det_fill."""
    import json
    from oracle import vqvae_ref
    surf = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "surface.json")))
    return {k: vqvae_ref.det_fill(k, s) for k, s, *_ in surf["vqvae_d"]}


def test_discriminators_and_losses_restatement(golden_dir):
    """oracle/vqvae_ref.py (DiscriminatorS/P, MPD, losses) vs the reference-generated fixture vqvae_disc.npz."""
    from oracle import vqvae_ref
    g = np.load(os.path.join(golden_dir, "vqvae_disc.npz"))
    sd = {k: v.requires_grad_(True) for k, v in _mpd_state_dict().items()}
    y = torch.from_numpy(g["y"]); y_hat = torch.from_numpy(g["y_hat"]).requires_grad_(True)
    dr, dg, _, _ = vqvae_ref.mpd_forward(sd, y, y_hat.detach())
    loss_d = vqvae_ref.discriminator_loss(dr, dg)
    np.testing.assert_allclose(loss_d.item(), g["loss_disc"], rtol=1e-5)
    for i in range(6):
        np.testing.assert_allclose(dr[i].detach().numpy(), g[f"logit_r{i}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(dg[i].detach().numpy(), g[f"logit_g{i}"], rtol=1e-4, atol=1e-5)
    loss_d.backward()
    np.testing.assert_allclose([v.grad.abs().sum().item() for v in sd.values()], g["d_grad_abs_sum"], rtol=2e-4)
    dr, dg, fr, fg = vqvae_ref.mpd_forward(sd, y, y_hat)
    lfm, lgen = vqvae_ref.feature_loss(fr, fg), vqvae_ref.generator_loss(dg)
    np.testing.assert_allclose([lfm.item(), lgen.item()], [g["loss_fm"], g["loss_gen"]], rtol=1e-5)
    (lfm + lgen).backward()
    np.testing.assert_allclose(y_hat.grad.numpy(), g["dy_hat"], rtol=1e-3, atol=1e-6)
    zs = [torch.from_numpy(a).requires_grad_(True) for a in g["kl_in"]]
    kl = vqvae_ref.kl_loss(*zs, torch.from_numpy(g["kl_mask"]))
    np.testing.assert_allclose(kl.item(), g["kl"], rtol=1e-6)
    kl.backward()
    np.testing.assert_allclose(np.stack([z.grad.numpy() for z in zs]), g["kl_grads"], rtol=1e-5, atol=1e-7)


def _flow_fixture(golden_dir):
    return np.load(os.path.join(golden_dir, "vqvae_flow.npz"))


def _sd_from(module_cls_keys):
    from oracle import vqvae_ref
    return {k: vqvae_ref.det_fill(k, s).requires_grad_(True) for k, s in module_cls_keys}


def test_wn_flow_snake_posterior_restatement(golden_dir):
    """oracle/vqvae_ref.py (WN, coupling block, anti-aliased SnakeBeta, PosteriorAudioEncoder) vs vqvae_flow.npz."""
    import json
    from oracle import vqvae_ref
    g = _flow_fixture(golden_dir)
    T = torch.from_numpy
    # WN
    keys = [(k[len("wn_grad:"):], g[k].shape) for k in g.files if k.startswith("wn_grad:")]
    sd = _sd_from(keys)
    x = T(g["wn_x"]).requires_grad_(True); gg = T(g["wn_g"]).requires_grad_(True)
    y = vqvae_ref.wn_forward(x, T(g["wn_mask"]), sd, "", 16, 5, 2, 3, gg)
    np.testing.assert_allclose(y.detach().numpy(), g["wn_y"], rtol=1e-5, atol=1e-5)
    (y * T(g["wn_ct"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["wn_dx"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gg.grad.numpy(), g["wn_dg"], rtol=1e-4, atol=1e-4)
    for k, _ in keys:
        np.testing.assert_allclose(sd[k].grad.numpy(), g["wn_grad:" + k], rtol=1e-4, atol=1e-4, err_msg=k)
    # coupling block
    keys = [(k[len("fl_grad:"):], g[k].shape) for k in g.files if k.startswith("fl_grad:")]
    sd = _sd_from(keys)
    x = T(g["fl_x"]).requires_grad_(True); gg = T(g["fl_g"]).requires_grad_(True)
    y = vqvae_ref.coupling_block_forward(x, T(g["wn_mask"]), sd, "", 8, 16, 5, 1, 2, 2, gg)
    np.testing.assert_allclose(y.detach().numpy(), g["fl_y"], rtol=1e-5, atol=1e-5)
    (y * T(g["fl_ct"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["fl_dx"], rtol=1e-4, atol=1e-4)
    # snake
    x = T(g["aa_x"]).requires_grad_(True); al = T(g["aa_alpha"]).requires_grad_(True); be = T(g["aa_beta"]).requires_grad_(True)
    np.testing.assert_allclose(vqvae_ref.kaiser_sinc_filter1d(0.25, 0.3, 12).numpy(), g["aa_fup"], rtol=1e-6)
    np.testing.assert_allclose(vqvae_ref.kaiser_sinc_filter1d(0.25, 0.3, 12).numpy(), g["aa_fdn"], rtol=1e-6)
    y = vqvae_ref.snake_aa(x, al, be)
    np.testing.assert_allclose(y.detach().numpy(), g["aa_y"], rtol=1e-5, atol=1e-6)
    (y * T(g["aa_ct"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["aa_dx"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(al.grad.numpy(), g["aa_dalpha"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(be.grad.numpy(), g["aa_dbeta"], rtol=1e-4, atol=1e-5)
    # posterior audio encoder
    keys = [(k, s) for k, s in json.loads(str(g["pe_keys"])) if not k.endswith("filter")]
    sd = _sd_from(keys)
    spec = T(g["pe_spec"]).requires_grad_(True); wav = T(g["pe_wav"]).requires_grad_(True); gg = T(g["pe_g"]).requires_grad_(True)
    z, m, logs = vqvae_ref.posterior_audio_encoder_forward(sd, "", spec, wav, T(g["pe_mask"]), gg, T(g["pe_noise"]))
    for a, k in ((z, "pe_z"), (m, "pe_m"), (logs, "pe_logs")):
        np.testing.assert_allclose(a.detach().numpy(), g[k], rtol=1e-4, atol=1e-4, err_msg=k)
    ct = T(g["pe_ct"])
    ((z * ct).sum() + 0.1 * (m * ct).sum() + 0.1 * logs.sum()).backward()
    np.testing.assert_allclose(wav.grad.numpy(), g["pe_dwav"], rtol=1e-3, atol=1e-4 * np.abs(g["pe_dwav"]).max())
    names = json.loads(str(g["pe_names"]))
    np.testing.assert_allclose([sd[k].grad.abs().sum().item() for k in names], g["pe_grad_abs_sum"], rtol=1e-3)


def test_attention_stacks_restatement(golden_dir):
    """oracle/vqvae_ref.py (Encoder / rel-pos MHA / FFN / MRTE / TextEncoder / MelStyleEncoder) vs vqvae_attn.npz."""
    import json
    from oracle import vqvae_ref
    g = np.load(os.path.join(golden_dir, "vqvae_attn.npz"))
    T = torch.from_numpy
    sd = _sd_from([(k, s) for k, s in json.loads(str(g["te_keys"]))])
    y = T(g["te_y"]).requires_grad_(True); ge = T(g["te_ge"]).requires_grad_(True)
    out, m, logs = vqvae_ref.text_encoder_forward(sd, "", y, T(g["te_ylen"]), T(g["te_text"]), T(g["te_tlen"]), ge)
    for a, k in ((out, "te_out"), (m, "te_m"), (logs, "te_logs")):
        np.testing.assert_allclose(a.detach().numpy(), g[k], rtol=1e-4, atol=2e-6, err_msg=k)
    ct = T(g["te_ct"])
    ((out * ct).sum() + (m * ct).sum() + 0.5 * logs.sum()).backward()
    np.testing.assert_allclose(y.grad.numpy(), g["te_dy"], rtol=1e-3, atol=1e-5 * np.abs(g["te_dy"]).max())
    np.testing.assert_allclose(ge.grad.numpy(), g["te_dge"], rtol=1e-3, atol=1e-5 * np.abs(g["te_dge"]).max())
    names = json.loads(str(g["te_names"]))
    got = np.array([sd[k].grad.abs().sum().item() if sd[k].grad is not None else 0.0 for k in names])
    np.testing.assert_allclose(got, g["te_grad_abs_sum"], rtol=2e-3, atol=1e-6)
    sd = _sd_from([(k, s) for k, s in json.loads(str(g["se_keys"]))])
    x = T(g["se_x"]).requires_grad_(True); mask = T(g["se_mask"])
    w = vqvae_ref.mel_style_encoder_forward(sd, "", x * mask, mask)
    np.testing.assert_allclose(w.detach().numpy(), g["se_w"], rtol=1e-4, atol=1e-5)
    (w * T(g["se_ct"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["se_dx"], rtol=1e-3, atol=1e-5 * np.abs(g["se_dx"]).max())


def test_synthesizer_forward_restatement(golden_dir):
    """oracle.vqvae_ref.synthesizer_forward (the composition: style encoder, two posterior encoders, stride-2 projection,
    codebook, text encoder + MRTE, coupling flow, segment slice, HiFi-GAN decoder) vs the reference's full forward in
    tests/golden/vqvae_step.npz (78.6 M det_fill parameters, injected noise / segment starts)."""
    import json
    from oracle import mel_ref, vqvae_ref
    g = np.load(os.path.join(golden_dir, "vqvae_step.npz"))
    surf = json.load(open(os.path.join(golden_dir, "surface.json")))
    cfg = json.loads(str(g["cfg"]))
    T = torch.from_numpy
    sd = {k: vqvae_ref.det_fill(k, s, 0.4) for k, s, *_ in surf["vqvae_g"] if not k.startswith("quantizer.") and not k.endswith("filter")}
    embed = vqvae_ref.det_fill("codebook.embed", (1024, 192)) * 2.0
    buffers = {"embed": embed.clone(), "embed_avg": embed * 4.0, "cluster_size": torch.full((1024,), 4.0)}
    wav = T(g["wav"]); wav_lengths = T(g["wav_lengths"])
    spec = mel_ref.spectrogram(wav, 2048, 640, 2048)
    with torch.no_grad():
        o, commit, _, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q), quantized = vqvae_ref.synthesizer_forward(
            sd, cfg, buffers, wav, wav, wav_lengths, spec, spec, wav_lengths // 640, T(g["text"]), T(g["text_lengths"]),
            T(g["noise_p"]), T(g["noise_q"]), T(g["ids_slice"]))
    for a, k, tol in ((z, "z", 1e-4), (m_q, "m_q", 1e-4), (logs_q, "logs_q", 1e-4), (quantized, "quantized", 1e-4),
                      (m_p, "m_p", 3e-4), (logs_p, "logs_p", 3e-4), (z_p, "z_p", 3e-4), (o, "o", 1e-3)):
        err = np.abs(a.numpy() - g[k]).max()
        assert err <= tol * np.abs(g[k]).max() + 1e-6, (k, err, np.abs(g[k]).max())
    np.testing.assert_allclose(commit.item(), g["commit"], rtol=1e-4)
    np.testing.assert_allclose(buffers["cluster_size"].numpy(), g["cb_cluster_size"], rtol=1e-5)


def test_synthesizer_infer_and_decode_restatement(golden_dir):
    """oracle.vqvae_ref.synthesizer_infer / synthesizer_decode (reverse coupling flow + whole-clip HiFi-GAN decoding) vs the
    reference's SynthesizerTrn.infer and the intended composition of its .decode (tests/golden/vqvae_infer.npz,
    tools/make_goldens.py gen_vq_infer): sub-sampled waveform, its head, and the absolute sum."""
    import json
    from oracle import mel_ref, vqvae_ref
    st = np.load(os.path.join(golden_dir, "vqvae_step.npz"))
    g = np.load(os.path.join(golden_dir, "vqvae_infer.npz"))
    surf = json.load(open(os.path.join(golden_dir, "surface.json")))
    cfg = json.loads(str(st["cfg"]))
    T = torch.from_numpy
    sd = {k: vqvae_ref.det_fill(k, s, 0.4) for k, s, *_ in surf["vqvae_g"] if not k.startswith("quantizer.") and not k.endswith("filter")}
    embed = vqvae_ref.det_fill("codebook.embed", (1024, 192)) * 2.0
    buffers = {"embed": embed.clone(), "embed_avg": embed * 4.0, "cluster_size": torch.full((1024,), 4.0)}
    wav, wav_lengths = T(st["wav"]), T(st["wav_lengths"])
    spec = mel_ref.spectrogram(wav, 2048, 640, 2048)
    with torch.no_grad():
        o = vqvae_ref.synthesizer_infer(sd, cfg, buffers, wav, wav_lengths, spec, wav_lengths // 640, T(st["text"]), T(st["text_lengths"]),
                                        T(g["noise_p"]), T(g["noise"]))
        od = vqvae_ref.synthesizer_decode(sd, cfg, buffers, T(g["dec_codes"]), T(st["text"])[:1, :int(st["text_lengths"][0])], spec[:1],
                                          T(g["dec_noise"]))
    scale = np.abs(g["o_head"]).max()
    assert tuple(o.shape) == (2, 1, 32000) and od.shape[-1] == int(g["dec_len"][0])
    assert np.abs(o.numpy()[:, :, ::8] - g["o_sub8"]).max() <= 1e-3 * scale and np.abs(o.numpy()[:, :, :2048] - g["o_head"]).max() <= 1e-3 * scale
    np.testing.assert_allclose(float(o.abs().sum()), g["o_abs_sum"][0], rtol=1e-3)
    assert np.abs(od.numpy()[:, :, :4096] - g["dec_o_head"]).max() <= 1e-3 * np.abs(g["dec_o_head"]).max()
    np.testing.assert_allclose(float(od.abs().sum()), g["dec_o_abs_sum"][0], rtol=1e-3)


def kmeans_case(g, name):
    seed, N, Kc = (int(v) for v in g[name + ":x_seed_N_K"])
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, D), dtype=np.float32)
    x[: N // 2] += rng.standard_normal((1, D), dtype=np.float32) * 2.0
    return torch.from_numpy(x), Kc, [torch.from_numpy(d) for d in g[name + ":draws"]]


def test_kmeans_init_and_dead_code_expiry(golden_dir):
    """First training batch of an un-initialised codebook (core_vq.py:141-168,205-230), random index draws injected."""
    g = np.load(os.path.join(golden_dir, "vq.npz"))
    for name in ("kmeans_perm", "kmeans_randint"):
        x, Kc, draws = kmeans_case(g, name)
        assert int(g[name + ":n_draws_used"]) == 2          # the k-means seed AND the expiry draw were consumed
        buf = {"embed": torch.zeros(Kc, D), "embed_avg": torch.zeros(Kc, D), "cluster_size": torch.zeros(Kc),
               "inited": torch.zeros(1)}
        q, ind = vq_ref.codebook_forward(x, buf, True, kmeans_iters=4, draws=draws)
        assert np.array_equal(ind.numpy(), g[name + ":ind"]), name
        np.testing.assert_allclose(q.numpy(), g[name + ":quantize"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(buf["cluster_size"].numpy(), g[name + ":cluster_size"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(buf["embed_avg"].numpy(), g[name + ":embed_avg"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(buf["embed"].numpy(), g[name + ":embed"], rtol=1e-5, atol=1e-5)
        assert float(buf["inited"]) == float(g[name + ":inited"][0]) == 1.0


def test_gan_step_restatement_losses_and_grad_norms(golden_dir):
    """oracle.vqvae_ref.gan_step_losses (the two-phase step body, train.py:313-406) vs the reference's own step in
    tests/golden/vqvae_step.npz: six losses, and both gradient 2-norms through CPU autograd."""
    import json
    from oracle import vqvae_ref
    g = np.load(os.path.join(golden_dir, "vqvae_step.npz"))
    surf = json.load(open(os.path.join(golden_dir, "surface.json")))
    cfg = json.loads(str(g["cfg"])); hps = json.loads(str(g["hps"]))
    T = torch.from_numpy
    params_g = set(json.loads(str(g["g_names"])))
    sd_g = {k: vqvae_ref.det_fill(k, s, 0.4) for k, s, *_ in surf["vqvae_g"] if not k.startswith("quantizer.") and not k.endswith("filter")}
    for k in sd_g:
        if k in params_g:
            sd_g[k].requires_grad_(True)
    sd_d = {k: vqvae_ref.det_fill(k, s, 0.6).requires_grad_(True) for k, s, *_ in surf["vqvae_d"]}
    embed = vqvae_ref.det_fill("codebook.embed", (1024, 192)) * 2.0
    buffers = {"embed": embed.clone(), "embed_avg": embed * 4.0, "cluster_size": torch.full((1024,), 4.0)}
    opt_d = torch.optim.AdamW(list(sd_d.values()), hps["learning_rate"], betas=hps["betas"], eps=hps["eps"])
    norms = {}

    def d_phase(loss_disc):                      # train.py:365-369: backward, norm, optimizer step -- BEFORE the generator phase
        loss_disc.backward()
        norms["d"] = sum(float(p.grad.double().pow(2).sum()) for p in sd_d.values()) ** 0.5
        opt_d.step()
        opt_d.zero_grad()
    ld, lg, terms = vqvae_ref.gan_step_losses(sd_g, sd_d, cfg, hps, buffers, T(g["wav"]), T(g["wav_lengths"]), T(g["text"]),
                                              T(g["text_lengths"]), T(g["noise_p"]), T(g["noise_q"]), T(g["ids_slice"]), d_update=d_phase)
    got = np.array([terms[k].item() for k in ("loss_disc", "loss_gen", "loss_fm", "loss_mel", "kl_ssl", "loss_kl")])
    np.testing.assert_allclose(got, g["losses"], rtol=2e-4)
    nd = norms["d"]
    lg.backward()
    ng = sum(float(p.grad.double().pow(2).sum()) for p in sd_g.values() if p.grad is not None) ** 0.5
    np.testing.assert_allclose([nd, ng], g["grad_norms"], rtol=2e-3)
