#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (/root/reference).

Runs only in the build container (the reference is not present on the GPU box); the fixtures it
writes are data (inputs as seeds / small arrays, expected outputs) and are committed.  Re-run with
    python tools/make_goldens.py            # all
    python tools/make_goldens.py gpt vq mel # subsets
Inputs and parameters that are too big to store are regenerated deterministically on both sides with
`oracle.gpt_ref.det_fill` / numpy `default_rng(seed)` (seeds are stored in the fixture).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402

librosa_mel_stub = ref_import.install()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import gpt_ref  # noqa: E402  (only for det_fill / synthetic inputs -- not for expected outputs)
from oracle import mel_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)

TINY_GPT = {"model_dim": 64, "max_mel_tokens": 64, "max_text_tokens": 32, "heads": 2,
            "use_mel_codes_as_input": True, "layers": 2, "number_text_tokens": 256,
            "number_mel_codes": 1026, "start_mel_token": 1024, "stop_mel_token": 1025,
            "start_text_token": 255, "train_solo_embeddings": False}


def sample(t, n=4096):
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride].numpy().copy()


def ref_gpt(cfg):
    import ttts.gpt.model as gm
    m = gm.UnifiedVoice(**cfg)
    sd = gpt_ref.det_state_dict(cfg)
    missing = m.load_state_dict(sd, strict=True)
    return m, sd


def tiny_inputs():
    g = torch.Generator().manual_seed(7)
    text = torch.randint(1, 255, (2, 12), generator=g, dtype=torch.int64)
    mel = torch.randint(0, 1024, (2, 24), generator=g, dtype=torch.int64)
    text_lengths = torch.tensor([12, 9])
    wav_lengths = torch.tensor([24 * 1024, 17 * 1024 + 5])  # second sample: padding rewritten to STOP
    text[1, 9:] = 0
    mel[1, 17:] = 0
    return text, text_lengths, mel, wav_lengths


def gen_gpt():
    import ttts.gpt.model as gm
    import ttts.vqvae.vq2 as vq2
    # ---- G0 surface -------------------------------------------------------------------------------
    cfg = json.load(open("/root/reference/ttts/gpt/config.json"))["gpt"]
    m = gm.UnifiedVoice(**cfg)
    surface = {"gpt": [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]}
    vcfg = json.load(open("/root/reference/ttts/vqvae/config.json"))
    g = vq2.SynthesizerTrn(vcfg["data"]["filter_length"] // 2 + 1,
                           vcfg["train"]["segment_size"] // vcfg["data"]["hop_length"], **vcfg["vqvae"])
    d = vq2.MultiPeriodDiscriminator()
    surface["vqvae_g"] = [[k, list(v.shape), str(v.dtype)] for k, v in g.state_dict().items()]
    surface["vqvae_d"] = [[k, list(v.shape), str(v.dtype)] for k, v in d.state_dict().items()]
    surface["gpt_config"] = cfg
    surface["vqvae_config"] = vcfg
    json.dump(surface, open(os.path.join(OUT, "surface.json"), "w"))
    print("G0 surface:", len(surface["gpt"]), len(surface["vqvae_g"]), len(surface["vqvae_d"]))

    # ---- G1 tiny (eval: dropouts off), unequal lengths --------------------------------------------
    m, sd = ref_gpt(TINY_GPT)
    m.eval()
    text, tl, mel, wl = tiny_inputs()
    lt, lm, logits = m(text.clone(), tl, mel.clone(), wl)
    loss = lt * 0.01 + lm
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    gn = float(torch.sqrt(sum((gr.double() ** 2).sum() for gr in grads.values())))
    out = {"text": text.numpy(), "text_lengths": tl.numpy(), "mel": mel.numpy(), "wav_lengths": wl.numpy(),
           "loss_text": lt.detach().numpy(), "loss_mel": lm.detach().numpy(),
           "mel_logits": logits.detach().numpy(), "grad_norm": np.float64(gn),
           "cfg_json": np.array(json.dumps(TINY_GPT))}
    for k, gr in grads.items():
        out["grad:" + k] = sample(gr)
    np.savez_compressed(os.path.join(OUT, "gpt_tiny.npz"), **out)
    print("G1 tiny: loss_text %.6f loss_mel %.6f gradnorm %.6f" % (float(lt), float(lm), gn))

    # ---- G9 three optimizer steps on the tiny model (eval-mode forward; AdamW + LambdaLR + clip) ----
    m, sd = ref_gpt(TINY_GPT)
    m.eval()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.96), weight_decay=0.01)

    def warmup(step):  # ttts/gpt/train.py:36-40
        return float(step / 500) if step < 500 else 1
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=warmup)
    rec = {"cfg_json": np.array(json.dumps(TINY_GPT))}
    norms, losses = [], []
    for s in range(5):
        lt, lm, _ = m(text.clone(), tl, mel.clone(), wl)
        loss = lt * 0.01 + lm
        loss.backward()
        tn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        sched.step()
        norms.append(float(tn))
        losses.append(float(loss))
    rec["grad_norms"] = np.array(norms, np.float64)
    rec["losses"] = np.array(losses, np.float64)
    for k, p in m.named_parameters():
        rec["param:" + k] = sample(p, 1024)
        rec["delta:" + k] = sample(p.detach().double() - sd[k].double(), 1024).astype(np.float64)
        st = opt.state[p]
        rec["m:" + k] = sample(st["exp_avg"], 1024)
        rec["v:" + k] = sample(st["exp_avg_sq"], 1024)
    np.savez_compressed(os.path.join(OUT, "gpt_step.npz"), **rec)
    print("G9 step: norms", norms, "losses", losses)

    # ---- G1b full config, B=1 and the BASELINE shape (128 text + 1024 audio tokens) -----------------
    m, sd = ref_gpt(cfg)
    m.eval()
    text, tl, mel, wl = gpt_ref.synthetic_batch(B=1, seed=1234, cfg=cfg)
    lt, lm, logits = m(text.clone(), tl, mel.clone(), wl)
    (lt * 0.01 + lm).backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    gn = float(torch.sqrt(sum((gr.double() ** 2).sum() for gr in grads.values())))
    out = {"seed": np.int64(1234), "loss_text": lt.detach().numpy(), "loss_mel": lm.detach().numpy(),
           "logits_slice": logits.detach()[0, ::64, ::64].numpy(), "grad_norm": np.float64(gn)}
    for k in ["mel_head.weight", "gpt.h.0.attn.c_attn.weight", "gpt.h.5.mlp.c_fc.weight", "text_embedding.weight",
              "mel_pos_embedding.emb.weight", "gpt.h.3.ln_1.weight", "final_norm.bias"]:
        out["grad:" + k] = sample(grads[k], 2048)
    np.savez_compressed(os.path.join(OUT, "gpt_full_b1.npz"), **out)
    print("G1b full: loss_text %.6f loss_mel %.6f gradnorm %.6f" % (float(lt), float(lm), gn))


def gen_vq():
    import ttts.vqvae.core_vq as cv
    from ttts.vqvae.quantize import ResidualVectorQuantizer
    rec = {}
    K, D = 1024, 192
    cb = cv.EuclideanCodebook(dim=D, codebook_size=K, kmeans_init=True, kmeans_iters=50, threshold_ema_dead_code=2)
    cb.inited.fill_(1)
    # (a) random data at three scales, N = 1024 and the BASELINE N = 4096
    cases = [("n1024_s1", 1024, 1.0, 11), ("n1024_s01", 1024, 0.1, 12), ("n1024_s10", 1024, 10.0, 13),
             ("n4096_s1", 4096, 1.0, 14)]
    for name, N, s, seed in cases:
        rng = np.random.default_rng(seed)
        x = torch.from_numpy(rng.standard_normal((N, D), dtype=np.float32) * np.float32(s))
        e = torch.from_numpy(rng.standard_normal((K, D), dtype=np.float32))
        cb.embed.copy_(e)
        idx = cb.quantize(x)
        rec[name + ":seed_N_scale"] = np.array([seed, N, s], np.float64)
        rec[name + ":idx"] = idx.numpy().astype(np.int64)
    # (b) exact ties: duplicated codebook rows -> lowest index wins (torch.max on ties)
    rng = np.random.default_rng(21)
    e = rng.standard_normal((K, D), dtype=np.float32)
    e[512:] = e[:512]  # every code appears twice
    x = e[rng.integers(0, 512, 256)] + rng.standard_normal((256, D), dtype=np.float32) * np.float32(0.01)
    cb.embed.copy_(torch.from_numpy(e))
    rec["ties:idx"] = cb.quantize(torch.from_numpy(x)).numpy().astype(np.int64)
    rec["ties:seed"] = np.int64(21)
    assert int(rec["ties:idx"].max()) < 512
    # (c) train-mode VectorQuantization.forward through the RVQ wrapper (n_q=1), buffers pre-initialised
    rvq = ResidualVectorQuantizer(dimension=D, n_q=1, bins=K)
    rvq.train()
    code = rvq.vq.layers[0]._codebook
    rng = np.random.default_rng(31)
    e = rng.standard_normal((K, D), dtype=np.float32)
    x = rng.standard_normal((4, D, 128), dtype=np.float32)
    code.inited.fill_(1)
    code.embed.copy_(torch.from_numpy(e))
    code.embed_avg.copy_(torch.from_numpy(e) * 4.0)
    code.cluster_size.fill_(4.0)
    xt = torch.from_numpy(x).requires_grad_(True)
    q, codes, commit, qlist = rvq(xt, layers=[0])
    (q.sum() * 0.5 + commit).backward()
    rec["train:seed"] = np.int64(31)
    rec["train:quantized"] = q.detach().numpy()
    rec["train:codes"] = codes.numpy().astype(np.int64)
    rec["train:commit"] = commit.detach().numpy()
    rec["train:dx"] = xt.grad.numpy()
    rec["train:cluster_size"] = code.cluster_size.numpy().copy()
    rec["train:embed_avg_sample"] = sample(code.embed_avg, 8192)
    rec["train:embed_sample"] = sample(code.embed, 8192)
    # (d) first training batch of an un-initialised codebook: k-means init (core_vq.py:71-93,141-148), then the search, then
    #     dead-code expiry (:152-168), EMA and normalisation (:216-228).  The only random draws are the index vectors of
    #     `sample_vectors` (randperm[:num] when there are enough samples, randint otherwise): they are injected.
    for name, Kc, N, seed in (("kmeans_perm", 64, 300, 41), ("kmeans_randint", 512, 300, 42)):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((N, D), dtype=np.float32)
        x[: N // 2] += rng.standard_normal((1, D), dtype=np.float32) * 2.0        # two loose clouds
        draws = [rng.permutation(N)[:Kc] if N >= Kc else rng.integers(0, N, Kc), rng.permutation(N)[:Kc] if N >= Kc else rng.integers(0, N, Kc)]
        queue = [torch.from_numpy(d.astype(np.int64)) for d in draws]
        used = []

        def injected(samples, num, _q=queue, _u=used):
            idx = _q.pop(0)
            assert idx.numel() == num and int(idx.max()) < samples.shape[0]
            _u.append(num)
            return samples[idx]
        cbk = cv.EuclideanCodebook(dim=D, codebook_size=Kc, kmeans_init=True, kmeans_iters=4, threshold_ema_dead_code=2)
        cbk.train()
        orig = cv.sample_vectors
        cv.sample_vectors = injected
        try:
            qz, ind = cbk(torch.from_numpy(x))
        finally:
            cv.sample_vectors = orig
        rec[name + ":x_seed_N_K"] = np.array([seed, N, Kc], np.int64)
        rec[name + ":draws"] = np.stack(draws).astype(np.int64)
        rec[name + ":n_draws_used"] = np.int64(len(used))
        rec[name + ":ind"] = ind.numpy().astype(np.int64)
        rec[name + ":quantize"] = qz.numpy()
        rec[name + ":cluster_size"] = cbk.cluster_size.numpy().copy()
        rec[name + ":embed_avg"] = cbk.embed_avg.numpy().copy()
        rec[name + ":embed"] = cbk.embed.numpy().copy()
        rec[name + ":inited"] = cbk.inited.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "vq.npz"), **rec)
    print("G3 vq:", {k: v.shape for k, v in rec.items() if k.endswith("idx")}, "commit", float(commit))


def gen_mel():
    import ttts.utils.data_utils as du
    rec = {}
    # mel-basis provenance check: our Slaney restatement vs the transformers implementation behind the stub
    for (sr, n_fft, n_mels, fmin, fmax) in [(32000, 2048, 128, 0, None), (22050, 1024, 80, 0, 8000)]:
        a = librosa_mel_stub(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax)
        b = mel_ref.slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax)
        err = np.abs(a - b).max()
        print("mel basis %d/%d/%d max|diff| = %.3e (max %.3e)" % (sr, n_fft, n_mels, err, np.abs(a).max()))
        assert err < 1e-6 * max(1.0, np.abs(a).max()) + 1e-7
    rng = np.random.default_rng(41)
    # set A: the vqvae config (32 kHz, 2048/640/2048, 128 mels, fmin 0, fmax null)  vqvae/config.json:53-64
    t = np.arange(32000) / 32000.0
    wav = np.stack([0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3300 * t + 1.0),
                    0.3 * np.sin(2 * np.pi * 700 * t * (1 + 0.2 * t))]).astype(np.float32)
    wav = np.clip(wav + 0.05 * rng.standard_normal(wav.shape, dtype=np.float32), -1, 1)
    y = torch.from_numpy(wav)
    spec = du.spectrogram_torch(y, 2048, 640, 2048, center=False)
    mel = du.spec_to_mel_torch(spec, 2048, 128, 32000, 0, None)
    rec["A:wav"] = wav
    rec["A:spec"] = spec.numpy()
    rec["A:mel"] = mel.numpy()
    # differentiable path on a segment (train.py:362-371,394): d/dy  mean|mel(y) - target| * 45
    yseg = torch.from_numpy(wav[:, :20480].copy()).requires_grad_(True)
    target = torch.from_numpy(rng.standard_normal((2, 128, 32), dtype=np.float32))
    yh_mel = du.mel_spectrogram_torch(yseg, 2048, 128, 32000, 640, 2048, 0, None)
    loss = torch.nn.functional.l1_loss(yh_mel, target) * 45
    loss.backward()
    rec["A:seg_target"] = target.numpy()
    rec["A:seg_mel"] = yh_mel.detach().numpy()
    rec["A:seg_loss"] = loss.detach().numpy()
    rec["A:seg_grad"] = yseg.grad.numpy()
    # set B: 22.05 kHz, 1024/256/1024, 80 mels, fmax 8000 (utils/utils.py:388-389 style front-end)
    mel_basis_cache = du.mel_basis
    mel_basis_cache.clear()
    wavb = np.clip(0.5 * rng.standard_normal((2, 22050), dtype=np.float32), -1, 1)
    yb = torch.from_numpy(wavb)
    specb = du.spectrogram_torch(yb, 1024, 256, 1024, center=False)
    melb = du.mel_spectrogram_torch(yb, 1024, 80, 22050, 256, 1024, 0, 8000)
    rec["B:wav"] = wavb
    rec["B:spec"] = specb.numpy()
    rec["B:mel"] = melb.numpy()
    np.savez_compressed(os.path.join(OUT, "mel.npz"), **rec)
    print("G4 mel:", spec.shape, mel.shape, specb.shape, melb.shape, "seg loss", float(loss))


TINY_GEN = {"initial_channel": 16, "resblock": "1", "resblock_kernel_sizes": [3, 7, 11],
            "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "upsample_rates": [4, 2],
            "upsample_initial_channel": 32, "upsample_kernel_sizes": [8, 4], "gin_channels": 8}


def gen_vqvae():
    """G5: tiny HiFi-GAN Generator (both weight-norm styles, ConvTranspose ups, ResBlock1 stacks): forward + grads."""
    import ttts.vqvae.vq2 as vq2
    torch.manual_seed(0)
    gen = vq2.Generator(**TINY_GEN)
    rng = np.random.default_rng(51)
    with torch.no_grad():
        for k, p in gen.named_parameters():
            a = rng.standard_normal(tuple(p.shape), dtype=np.float32)
            if k.endswith("weight_g") or k.endswith("original0"):
                a = 0.5 + 0.1 * np.abs(a)
            elif p.dim() > 1:
                a = a * (0.3 / np.sqrt(p.shape[1] * p.shape[2]))
            else:
                a = a * 0.05
            p.copy_(torch.from_numpy(a))
    x = torch.from_numpy(rng.standard_normal((2, 16, 20), dtype=np.float32)).requires_grad_(True)
    g = torch.from_numpy(rng.standard_normal((2, 8, 1), dtype=np.float32)).requires_grad_(True)
    ct = torch.from_numpy(rng.standard_normal((2, 1, 160), dtype=np.float32))
    y = gen(x, g)
    (y * ct).sum().backward()
    rec = {"cfg_json": np.array(json.dumps(TINY_GEN)), "x": x.detach().numpy(), "g": g.detach().numpy(), "ct": ct.numpy(),
           "y": y.detach().numpy(), "dx": x.grad.numpy(), "dg": g.grad.numpy()}
    for k, v in gen.state_dict().items():
        rec["sd:" + k] = v.numpy()
    for k, p in gen.named_parameters():
        rec["grad:" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "vqvae_generator.npz"), **rec)
    print("G5 generator:", y.shape, "params", sum(p.numel() for p in gen.parameters()), "keys", len(gen.state_dict()))


def gen_disc():
    """G5b/G7: the reference MultiPeriodDiscriminator + losses.py on a 2 x 4100-sample pair (4100 is not a multiple of
    any period: exercises the reflect pad).  Weights: oracle.vqvae_ref.det_fill (not stored)."""
    import ttts.vqvae.vq2 as vq2
    import ttts.vqvae.losses as L
    from oracle import vqvae_ref
    torch.manual_seed(0)
    mpd = vq2.MultiPeriodDiscriminator()
    with torch.no_grad():
        for k, p in mpd.named_parameters():
            p.copy_(vqvae_ref.det_fill(k, p.shape))
    rng = np.random.default_rng(77)
    y = torch.from_numpy((rng.standard_normal((2, 1, 4100)) * 0.3).astype(np.float32))
    y_hat = torch.from_numpy((rng.standard_normal((2, 1, 4100)) * 0.3).astype(np.float32)).requires_grad_(True)
    rec = {"y": y.numpy(), "y_hat": y_hat.detach().numpy()}
    # discriminator phase
    dr, dg, _, _ = mpd(y, y_hat.detach())
    loss_d, r_l, g_l = L.discriminator_loss(dr, dg)
    mpd.zero_grad()
    loss_d.backward()
    rec["loss_disc"] = loss_d.detach().numpy(); rec["r_losses"] = np.array(r_l, np.float32); rec["g_losses"] = np.array(g_l, np.float32)
    rec["d_grad_abs_sum"] = np.array([p.grad.abs().sum().item() for _, p in mpd.named_parameters()], np.float64)
    rec["d_grad_sum"] = np.array([p.grad.sum().item() for _, p in mpd.named_parameters()], np.float64)
    for i in range(6):
        rec[f"logit_r{i}"] = dr[i].detach().numpy(); rec[f"logit_g{i}"] = dg[i].detach().numpy()
    # generator phase
    mpd.zero_grad()
    dr, dg, fr, fg = mpd(y, y_hat)
    loss_fm = L.feature_loss(fr, fg)
    loss_gen, _ = L.generator_loss(dg)
    (loss_fm + loss_gen).backward()
    rec["loss_fm"] = loss_fm.detach().numpy(); rec["loss_gen"] = loss_gen.detach().numpy()
    rec["dy_hat"] = y_hat.grad.numpy()
    rec["fmap_abs_mean"] = np.array([[f.abs().mean().item() for f in fl] + [0.0] * (7 - len(fl)) for fl in fg], np.float64)
    rec["fmap_shapes"] = np.array(json.dumps([[list(f.shape) for f in fl] for fl in fg]))
    # kl loss (G7)
    zs = [torch.from_numpy(rng.standard_normal((2, 6, 37)).astype(np.float32)).requires_grad_(True) for _ in range(4)]
    mask = torch.ones(2, 1, 37); mask[1, :, 25:] = 0
    kl = L.kl_loss(zs[0], zs[1], zs[2], zs[3], mask)
    kl.backward()
    rec["kl_in"] = np.stack([z.detach().numpy() for z in zs]); rec["kl_mask"] = mask.numpy(); rec["kl"] = kl.detach().numpy()
    rec["kl_grads"] = np.stack([z.grad.numpy() for z in zs])
    np.savez_compressed(os.path.join(OUT, "vqvae_disc.npz"), **rec)
    print("G5b disc:", float(loss_d), float(loss_fm), float(loss_gen), float(kl))


def _fill_det(module, gain=1.0):
    from oracle import vqvae_ref
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(vqvae_ref.det_fill(k, p.shape, gain))


def _grads(module):
    return {("grad:" + k): p.grad.numpy() for k, p in module.named_parameters() if p.grad is not None}


def gen_flow():
    """G5c: reference WN, ResidualCouplingBlock, Activation1d(SnakeBeta), PosteriorAudioEncoder (tiny / short inputs)."""
    import ttts.vqvae.modules as M
    import ttts.vqvae.vq2 as vq2
    from ttts.vqvae import activations
    from ttts.vqvae.alias_free_torch import Activation1d
    rng = np.random.default_rng(99)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    rec = {}
    # WN (dilation_rate 2 exercises the dilated in_layers)
    wn = M.WN(16, 5, 2, 3, gin_channels=8)
    _fill_det(wn)
    x = t(2, 16, 50).requires_grad_(True); g = t(2, 8, 1).requires_grad_(True)
    mask = torch.ones(2, 1, 50); mask[1, :, 37:] = 0
    ct = t(2, 16, 50)
    y = wn(x, mask, g=g)
    (y * ct).sum().backward()
    rec.update({"wn_x": x.detach().numpy(), "wn_g": g.detach().numpy(), "wn_mask": mask.numpy(), "wn_ct": ct.numpy(),
                "wn_y": y.detach().numpy(), "wn_dx": x.grad.numpy(), "wn_dg": g.grad.numpy()})
    rec.update({"wn_" + k: v for k, v in _grads(wn).items()})
    # coupling block
    fl = vq2.ResidualCouplingBlock(8, 16, 5, 1, 2, n_flows=2, gin_channels=8)
    _fill_det(fl)
    x = t(2, 8, 50).requires_grad_(True); g = t(2, 8, 1).requires_grad_(True); ct = t(2, 8, 50)
    y = fl(x, mask, g=g)
    (y * ct).sum().backward()
    rec.update({"fl_x": x.detach().numpy(), "fl_g": g.detach().numpy(), "fl_ct": ct.numpy(), "fl_y": y.detach().numpy(),
                "fl_dx": x.grad.numpy(), "fl_dg": g.grad.numpy()})
    rec.update({"fl_" + k: v for k, v in _grads(fl).items()})
    # anti-aliased snake
    act = Activation1d(activation=activations.SnakeBeta(6, alpha_logscale=True))
    with torch.no_grad():
        act.act.alpha.copy_(t(6) * 0.3); act.act.beta.copy_(t(6) * 0.3)
    x = t(2, 6, 40).requires_grad_(True); ct = t(2, 6, 40)
    y = act(x)
    (y * ct).sum().backward()
    rec.update({"aa_x": x.detach().numpy(), "aa_ct": ct.numpy(), "aa_alpha": act.act.alpha.detach().numpy(),
                "aa_beta": act.act.beta.detach().numpy(), "aa_y": y.detach().numpy(), "aa_dx": x.grad.numpy(),
                "aa_dalpha": act.act.alpha.grad.numpy(), "aa_dbeta": act.act.beta.grad.numpy(),
                "aa_fup": act.upsample.filter.numpy(), "aa_fdn": act.downsample.lowpass.filter.numpy()})
    # PosteriorAudioEncoder: spec 20 channels, T = 8 frames (wav 5120), gin 16; weights det_fill (not stored)
    enc = vq2.PosteriorAudioEncoder(20, 192, 192, 5, 1, 16, gin_channels=16)
    _fill_det(enc)
    spec = t(2, 20, 8).requires_grad_(True); wav = (t(2, 1, 5120) * 0.3).requires_grad_(True); g = t(2, 16, 1).requires_grad_(True)
    mask = torch.ones(2, 1, 8); mask[1, :, 6:] = 0
    noise = t(2, 192, 8)
    orig = torch.randn_like
    torch.randn_like = lambda m_, *a, **k: noise
    try:
        z, m, logs = enc(spec, wav, mask, g=g)
    finally:
        torch.randn_like = orig
    ctz = t(2, 192, 8)
    ((z * ctz).sum() + 0.1 * (m * ctz).sum() + 0.1 * logs.sum()).backward()
    names = [k for k, _ in enc.named_parameters()]
    rec.update({"pe_spec": spec.detach().numpy(), "pe_wav": wav.detach().numpy(), "pe_g": g.detach().numpy(),
                "pe_mask": mask.numpy(), "pe_noise": noise.numpy(), "pe_ct": ctz.numpy(), "pe_z": z.detach().numpy(),
                "pe_m": m.detach().numpy(), "pe_logs": logs.detach().numpy(), "pe_dspec": spec.grad.numpy(),
                "pe_dwav": wav.grad.numpy(), "pe_dg": g.grad.numpy(),
                "pe_names": np.array(json.dumps(names)),
                "pe_grad_abs_sum": np.array([p.grad.abs().sum().item() for _, p in enc.named_parameters()], np.float64),
                "pe_grad_sum": np.array([p.grad.sum().item() for _, p in enc.named_parameters()], np.float64),
                "pe_keys": np.array(json.dumps([[k, list(v.shape)] for k, v in enc.state_dict().items()]))})
    np.savez_compressed(os.path.join(OUT, "vqvae_flow.npz"), **rec)
    print("G5c flow/wn/aa/posterior:", float(z.abs().mean()), len(names))


def gen_attn():
    """G6: reference TextEncoder (3 + 6 + 3 relative-attention layers, MRTE cross-attention) and MelStyleEncoder in eval
    mode on short ragged inputs; weights det_fill (not stored)."""
    import ttts.vqvae.vq2 as vq2
    import ttts.vqvae.modules as M
    rng = np.random.default_rng(123)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    rec = {}
    te = vq2.TextEncoder(192, 192, 768, 2, 6, 3, 0.1).eval()
    _fill_det(te)
    y = t(2, 192, 40).requires_grad_(True); ge = t(2, 512, 1).requires_grad_(True)
    y_len = torch.tensor([40, 29]); text = torch.from_numpy(rng.integers(1, 255, (2, 12))); text_len = torch.tensor([12, 7])
    out, m, logs = te(y, y_len, text, text_len, ge)
    ct = t(2, 192, 40)
    ((out * ct).sum() + (m * ct).sum() + 0.5 * logs.sum()).backward()
    names = [k for k, _ in te.named_parameters()]
    rec.update({"te_y": y.detach().numpy(), "te_ge": ge.detach().numpy(), "te_ylen": y_len.numpy(), "te_text": text.numpy(),
                "te_tlen": text_len.numpy(), "te_ct": ct.numpy(), "te_out": out.detach().numpy(), "te_m": m.detach().numpy(),
                "te_logs": logs.detach().numpy(), "te_dy": y.grad.numpy(), "te_dge": ge.grad.numpy(),
                "te_names": np.array(json.dumps(names)),
                "te_keys": np.array(json.dumps([[k, list(v.shape)] for k, v in te.state_dict().items()])),
                "te_grad_abs_sum": np.array([p.grad.abs().sum().item() if p.grad is not None else 0.0 for _, p in te.named_parameters()]),
                "te_grad_sum": np.array([p.grad.sum().item() if p.grad is not None else 0.0 for _, p in te.named_parameters()])})
    se = M.MelStyleEncoder(40, style_vector_dim=512).eval()
    _fill_det(se)
    x = t(2, 40, 33).requires_grad_(True)
    mask = torch.ones(2, 1, 33); mask[1, :, 20:] = 0
    w = se(x * mask, mask)
    ctw = t(2, 512, 1)
    (w * ctw).sum().backward()
    names = [k for k, _ in se.named_parameters()]
    rec.update({"se_x": x.detach().numpy(), "se_mask": mask.numpy(), "se_ct": ctw.numpy(), "se_w": w.detach().numpy(),
                "se_dx": x.grad.numpy(), "se_names": np.array(json.dumps(names)),
                "se_keys": np.array(json.dumps([[k, list(v.shape)] for k, v in se.state_dict().items()])),
                "se_grad_abs_sum": np.array([p.grad.abs().sum().item() for _, p in se.named_parameters()]),
                "se_grad_sum": np.array([p.grad.sum().item() for _, p in se.named_parameters()])})
    np.savez_compressed(os.path.join(OUT, "vqvae_attn.npz"), **rec)
    print("G6 attn:", float(out.abs().mean()), float(w.abs().mean()))


STEP_HPS = {"filter_length": 2048, "hop_length": 640, "win_length": 2048, "n_mel_channels": 128, "sampling_rate": 32000,
            "mel_fmin": 0.0, "mel_fmax": None, "segment_size": 20480, "c_mel": 45, "c_kl": 1.0, "learning_rate": 1e-4,
            "betas": [0.8, 0.99], "eps": 1e-9}


STEP_GAIN = 0.4


def gen_step():
    """G8: one full two-phase VQ-VAE-GAN step of the reference (train.py:313-406 restated around the imported modules):
    SynthesizerTrn(full vqvae/config.json, p_dropout 0, ref_enc.eval()) + MultiPeriodDiscriminator on 2 clips of
    50 / 40 frames, injected noise and segment starts, pre-initialised codebook, AdamW(1e-4, (0.8, 0.99), 1e-9)."""
    import ttts.vqvae.vq2 as vq2
    import ttts.vqvae.losses as L
    import ttts.utils.commons as commons
    import ttts.utils.data_utils as du
    from oracle import vqvae_ref
    h = STEP_HPS
    cfg = json.load(open("/root/reference/ttts/vqvae/config.json"))["vqvae"]
    cfg["p_dropout"] = 0.0
    torch.manual_seed(0)
    net_g = vq2.SynthesizerTrn(h["filter_length"] // 2 + 1, h["segment_size"] // h["hop_length"], **cfg)
    net_d = vq2.MultiPeriodDiscriminator()
    _fill_det(net_g, STEP_GAIN); _fill_det(net_d, 0.6)
    net_g.train(); net_d.train(); net_g.ref_enc.eval()
    cb = net_g.quantizer.vq.layers[0]._codebook
    with torch.no_grad():
        cb.inited.fill_(1)
        cb.embed.copy_(vqvae_ref.det_fill("codebook.embed", cb.embed.shape) * 2.0)
        cb.embed_avg.copy_(cb.embed * 4.0)
        cb.cluster_size.fill_(4.0)
    rng = np.random.default_rng(2024)
    tt = np.arange(32000) / 32000.0
    wav = np.stack([sum(rng.uniform(0.02, 0.2) * np.sin(2 * np.pi * rng.uniform(60, 6000) * tt + rng.uniform(0, 6.28))
                        for _ in range(16)) for _ in range(2)]).astype(np.float32)
    wav = np.clip(wav + 0.05 * rng.standard_normal(wav.shape).astype(np.float32), -1, 1)
    wav[1, 25600:] = 0
    wav = torch.from_numpy(wav)
    wav_lengths = torch.tensor([32000, 25600])
    text = torch.from_numpy(rng.integers(1, 255, (2, 16))); text_lengths = torch.tensor([16, 11])
    noises = [torch.from_numpy(rng.standard_normal((2, 192, 50)).astype(np.float32)) for _ in range(2)]
    ids = torch.tensor([3, 5])
    calls = {"n": 0}

    def fake_randn_like(t_, *a, **k):
        calls["n"] += 1
        return noises[calls["n"] - 1]

    def fake_slice(x, x_lengths=None, segment_size=4):
        return commons.slice_segments(x, ids, segment_size), ids

    optim_g = torch.optim.AdamW(net_g.parameters(), h["learning_rate"], betas=h["betas"], eps=h["eps"])
    optim_d = torch.optim.AdamW(net_d.parameters(), h["learning_rate"], betas=h["betas"], eps=h["eps"])
    g_before = {k: p.detach().clone() for k, p in net_g.named_parameters()}
    d_before = {k: p.detach().clone() for k, p in net_d.named_parameters()}
    orig_randn_like, orig_slice = torch.randn_like, commons.rand_slice_segments
    # code extraction (vq2.py:912-919 is not runnable as written: undefined `y_lengths`, full-rate mask on the half-rate
    # projection -- SURVEY App. B): the same reference modules composed as the training forward composes them
    # (ref_enc -> enc_p -> proj -> quantizer), eval mode, posterior noise zero, on a copy taken BEFORE the step
    ng_eval = vq2.SynthesizerTrn(h["filter_length"] // 2 + 1, h["segment_size"] // h["hop_length"], **cfg)
    ng_eval.load_state_dict(net_g.state_dict())
    ng_eval.eval()
    torch.randn_like = lambda t_, *a, **k: torch.zeros_like(t_)
    try:
        with torch.no_grad():
            spec0 = du.spectrogram_torch(wav, h["filter_length"], h["hop_length"], h["win_length"], center=False).squeeze(0)
            sl0 = torch.LongTensor([x // h["hop_length"] for x in wav_lengths])
            m0 = torch.unsqueeze(commons.sequence_mask(sl0, spec0.size(2)), 1).to(spec0.dtype)
            ge0 = ng_eval.ref_enc(spec0 * m0, m0)
            x0, _, _ = ng_eval.enc_p(spec0, wav.unsqueeze(1), m0, g=ge0)
            _, latent_codes, _, _ = ng_eval.quantizer(ng_eval.proj(x0))
            latent_codes = latent_codes.transpose(0, 1).clone()
    finally:
        torch.randn_like = orig_randn_like
    box = {}
    def grab(mod, inp, out):          # a forward hook must return None, or its value replaces the module output
        box["codes"] = out[1].detach().clone()
    hook = net_g.quantizer.register_forward_hook(grab)
    torch.randn_like = fake_randn_like; commons.rand_slice_segments = fake_slice
    try:
        spec = du.spectrogram_torch(wav, h["filter_length"], h["hop_length"], h["win_length"], center=False).squeeze(0)
        spec_lengths = torch.LongTensor([x // h["hop_length"] for x in wav_lengths])
        y_hat, kl_ssl, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), quantized = net_g(
            wav, wav, wav_lengths, spec, spec, spec_lengths, text, text_lengths)
    finally:
        torch.randn_like = orig_randn_like; commons.rand_slice_segments = orig_slice
        hook.remove()
    assert calls["n"] == 2
    mel = du.spec_to_mel_torch(spec, h["filter_length"], h["n_mel_channels"], h["sampling_rate"], h["mel_fmin"], h["mel_fmax"])
    y_mel = commons.slice_segments(mel, ids_slice, h["segment_size"] // h["hop_length"])
    y_hat_mel = du.mel_spectrogram_torch(y_hat.squeeze(1), h["filter_length"], h["n_mel_channels"], h["sampling_rate"],
                                         h["hop_length"], h["win_length"], h["mel_fmin"], h["mel_fmax"])
    y = commons.slice_segments(wav.unsqueeze(1), ids_slice * h["hop_length"], h["segment_size"])
    y_d_hat_r, y_d_hat_g, _, _ = net_d(y, y_hat.detach())
    loss_disc, _, _ = L.discriminator_loss(y_d_hat_r, y_d_hat_g)
    optim_d.zero_grad(); loss_disc.backward()
    grad_norm_d = commons.clip_grad_value_(net_d.parameters(), None)
    d_grad_abs = np.array([p.grad.abs().sum().item() for _, p in net_d.named_parameters()])
    optim_d.step()
    y_d_hat_r, y_d_hat_g, fmap_r, fmap_g = net_d(y, y_hat)
    loss_mel = torch.nn.functional.l1_loss(y_mel, y_hat_mel) * h["c_mel"]
    loss_kl = L.kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * h["c_kl"]
    loss_fm = L.feature_loss(fmap_r, fmap_g)
    loss_gen, _ = L.generator_loss(y_d_hat_g)
    loss_gen_all = loss_gen + loss_fm + loss_mel + kl_ssl * 1 + loss_kl
    optim_g.zero_grad(); loss_gen_all.backward()
    grad_norm_g = commons.clip_grad_value_(net_g.parameters(), None)
    g_names = [k for k, _ in net_g.named_parameters()]
    g_grad_abs = np.array([p.grad.abs().sum().item() if p.grad is not None else -1.0 for _, p in net_g.named_parameters()])
    optim_g.step()
    rec = {"wav": wav.numpy(), "wav_lengths": wav_lengths.numpy(), "text": text.numpy(), "text_lengths": text_lengths.numpy(),
           "noise_p": noises[0].numpy(), "noise_q": noises[1].numpy(), "ids_slice": ids.numpy(),
           "o": y_hat.detach().numpy(), "commit": kl_ssl.detach().numpy(), "quantized": quantized.detach().numpy(),
           "z": z.detach().numpy(), "z_p": z_p.detach().numpy(), "m_p": m_p.detach().numpy(), "logs_p": logs_p.detach().numpy(),
           "m_q": m_q.detach().numpy(), "logs_q": logs_q.detach().numpy(), "y_mask": z_mask.numpy(),
           "losses": np.array([loss_disc.item(), loss_gen.item(), loss_fm.item(), loss_mel.item(), kl_ssl.item(), loss_kl.item()]),
           "grad_norms": np.array([grad_norm_d, grad_norm_g]), "d_grad_abs": d_grad_abs, "g_grad_abs": g_grad_abs,
           "g_names": np.array(json.dumps(g_names)),
           "g_delta_abs": np.array([(p.detach() - g_before[k]).abs().sum().item() for k, p in net_g.named_parameters()]),
           "d_delta_abs": np.array([(p.detach() - d_before[k]).abs().sum().item() for k, p in net_d.named_parameters()]),
           "cb_cluster_size": cb.cluster_size.numpy(), "cb_embed_avg_sum": cb.embed_avg.sum(1).numpy(),
           "cb_embed_head": cb.embed[:8].numpy(), "hps": np.array(json.dumps(h)), "cfg": np.array(json.dumps(cfg)),
           "codes": box["codes"].numpy().astype(np.int64), "latent_codes": latent_codes.numpy().astype(np.int64)}
    np.savez_compressed(os.path.join(OUT, "vqvae_step.npz"), **rec)
    print("G8 step losses:", rec["losses"], "norms", rec["grad_norms"], "unused g params:", int((g_grad_abs < 0).sum()))


def gen_vq_infer():
    """SynthesizerTrn.infer (vq2.py:873-889) and .decode (:891-910) of the reference on the clips of the G8 fixture, eval mode,
    injected noise.  `infer` runs as written.  `decode` is not runnable as written (undefined `text_legnths`, `y_mask`; y_lengths
    taken before the x2 upsampling -- SURVEY App. B): the same reference modules are composed the way its body intends
    (ref_enc -> quantizer.decode -> x2 nearest -> enc_p_2 -> sample -> reverse flow -> dec) with y_lengths = the upsampled length."""
    import ttts.vqvae.vq2 as vq2
    import ttts.utils.commons as commons
    import ttts.utils.data_utils as du
    from oracle import vqvae_ref
    h = STEP_HPS
    st = np.load(os.path.join(OUT, "vqvae_step.npz"))
    cfg = json.loads(str(st["cfg"]))
    torch.manual_seed(0)
    net_g = vq2.SynthesizerTrn(h["filter_length"] // 2 + 1, h["segment_size"] // h["hop_length"], **cfg)
    _fill_det(net_g, STEP_GAIN)
    net_g.eval()
    cb = net_g.quantizer.vq.layers[0]._codebook
    with torch.no_grad():
        cb.inited.fill_(1)
        cb.embed.copy_(vqvae_ref.det_fill("codebook.embed", cb.embed.shape) * 2.0)
    wav, wav_lengths = torch.from_numpy(st["wav"]), torch.from_numpy(st["wav_lengths"])
    text, text_lengths = torch.from_numpy(st["text"]), torch.from_numpy(st["text_lengths"])
    rng = np.random.default_rng(77)
    noises = [torch.from_numpy(rng.standard_normal((2, 192, 50)).astype(np.float32)) for _ in range(2)]
    calls = {"n": 0}

    def fake_randn_like(t_, *a, **k):
        calls["n"] += 1
        return noises[calls["n"] - 1]
    orig = torch.randn_like
    torch.randn_like = fake_randn_like
    try:
        with torch.no_grad():
            spec = du.spectrogram_torch(wav, h["filter_length"], h["hop_length"], h["win_length"], center=False).squeeze(0)
            spec_lengths = torch.LongTensor([x // h["hop_length"] for x in wav_lengths])
            o = net_g.infer(wav, wav_lengths, spec, spec_lengths, text, text_lengths, noise_scale=0.5)
    finally:
        torch.randn_like = orig
    assert calls["n"] == 2
    # decode: one clip (the reference builds its lengths from single tensors), codes of the first clip's extraction
    codes = torch.from_numpy(st["latent_codes"])[:1].transpose(0, 1).contiguous()   # (n_q, 1, T): the first clip
    refer = spec[:1]
    dn = torch.from_numpy(rng.standard_normal((1, 192, 2 * codes.size(2))).astype(np.float32))
    with torch.no_grad():
        refer_lengths = torch.LongTensor([refer.size(2)])
        refer_mask = torch.unsqueeze(commons.sequence_mask(refer_lengths, refer.size(2)), 1).to(refer.dtype)
        ge = net_g.ref_enc(refer * refer_mask, refer_mask)
        quantized = net_g.quantizer.decode(codes)
        quantized = torch.nn.functional.interpolate(quantized, size=int(quantized.shape[-1] * 2), mode="nearest")
        y_lengths = torch.LongTensor([quantized.size(2)])
        y_mask = torch.unsqueeze(commons.sequence_mask(y_lengths, quantized.size(2)), 1).to(refer.dtype)
        _, m_p, logs_p = net_g.enc_p_2(quantized, y_lengths, text[:1, :int(text_lengths[0])], text_lengths[:1], ge)
        z_p = m_p + dn * torch.exp(logs_p) * 0.5
        z = net_g.flow(z_p, y_mask, g=ge, reverse=True)
        od = net_g.dec(z * y_mask, g=ge)
    np.savez_compressed(os.path.join(OUT, "vqvae_infer.npz"), noise_p=noises[0].numpy(), noise=noises[1].numpy(),
                        o_sub8=o.numpy()[:, :, ::8].copy(), o_head=o.numpy()[:, :, :2048],
                        o_abs_sum=np.array([float(o.abs().sum())]), dec_noise=dn.numpy(), dec_codes=codes.numpy().astype(np.int64),
                        dec_o_head=od.numpy()[:, :, :4096], dec_o_abs_sum=np.array([float(od.abs().sum())]), dec_len=np.array([od.shape[-1]]))
    print("vq infer:", tuple(o.shape), float(o.abs().mean()), "decode:", tuple(od.shape), float(od.abs().mean()))


def gen_infer():
    """return_latent forward, and greedy decoding through the reference's GPT2InferenceModel.forward
    (kv_cache=False, as api_zh.py:52).  transformers 5.x removed GenerationMixin from PreTrainedModel, so
    `inference_model.generate` (model.py:559) cannot run here: the fixture drives inference_speech's own input assembly
    (model.py:536-547) and the cache-less forward step by step with HF's greedy rule (argmax, pad after eos), and pins
    the logits processors on seeded scores with the installed transformers classes + the reference's TypicalLogitsWarper."""
    from transformers.generation import logits_process as lp
    from ttts.utils.typical_sampling import TypicalLogitsWarper
    m, sd = ref_gpt(TINY_GPT)
    m.eval()
    m.post_init_gpt2_config(kv_cache=False)
    text, tl, mel, wl = tiny_inputs()
    rec = {"cfg_json": np.array(json.dumps(TINY_GPT))}
    with torch.no_grad():
        lat = m(text.clone(), tl, mel.clone(), wl, return_latent=True, clip_inputs=False)
        rec["latent"] = lat.numpy()
        # ---- greedy decode, B = 2, equal-length text (12) and a 6-code prompt
        g = torch.Generator().manual_seed(11)
        itext = torch.randint(1, 255, (2, 12), generator=g)
        prompt = torch.randint(0, 1024, (2, 6), generator=g)
        steps = 10
        t_in = F.pad(itext, (0, 1), value=m.stop_text_token)
        t_in, _ = m.build_aligned_inputs_and_targets(t_in, m.start_text_token, m.stop_text_token)
        emb = m.text_embedding(t_in) + m.text_pos_embedding(t_in)
        mel_in, _ = m.build_aligned_inputs_and_targets(prompt, m.start_mel_token, m.stop_mel_token)
        m.inference_model.store_mel_emb(emb)
        ids = torch.full((2, emb.shape[1] + mel_in.shape[-1]), 1, dtype=torch.long)
        ids[:, -mel_in.shape[1]:] = mel_in
        trunc = ids.shape[1]
        unfinished = torch.ones(2, dtype=torch.bool)
        raw = []
        for _ in range(steps):
            out = m.inference_model(input_ids=ids, attention_mask=torch.ones_like(ids), return_dict=True)
            logits = out.logits[:, -1].float()
            raw.append(logits.numpy())
            nxt = logits.argmax(-1)
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, m.stop_mel_token))
            ids = torch.cat([ids, nxt[:, None]], 1)
            unfinished &= nxt != m.stop_mel_token
        rec.update({"itext": itext.numpy(), "prompt": prompt.numpy(), "greedy_codes": ids[:, trunc:].numpy(),
                    "greedy_logits": np.stack(raw), "first_pass_logits": m.inference_model(
                        input_ids=ids[:, :trunc], attention_mask=torch.ones_like(ids[:, :trunc]), return_dict=True).logits.numpy()})
        margins = np.sort(np.stack(raw), -1)
        print("infer: greedy codes", ids[:, trunc:].tolist(), "min top-2 margin %.4f" % float((margins[..., -1] - margins[..., -2]).min()))
        # ---- logits processors on seeded scores (B 3, V 1026), history of 9 tokens
        scores = torch.randn(3, 1026, generator=g) * 3.0
        hist = torch.randint(0, 1026, (3, 9), generator=g)
        rec["proc_scores"], rec["proc_hist"] = scores.numpy(), hist.numpy()
        rec["proc_rep2"] = lp.RepetitionPenaltyLogitsProcessor(2.0)(hist, scores.clone()).numpy()
        rec["proc_temp08"] = lp.TemperatureLogitsWarper(0.8)(hist, scores.clone()).numpy()
        rec["proc_topk50"] = lp.TopKLogitsWarper(50)(hist, scores.clone()).numpy()
        rec["proc_topp08"] = lp.TopPLogitsWarper(0.8)(hist, scores.clone()).numpy()
        rec["proc_typical09"] = TypicalLogitsWarper(mass=0.9)(hist, scores.clone()).numpy()
        chain = lp.RepetitionPenaltyLogitsProcessor(2.0)(hist, scores.clone())
        chain = lp.TemperatureLogitsWarper(0.8)(hist, chain)
        chain = lp.TopKLogitsWarper(50)(hist, chain)
        chain = lp.TopPLogitsWarper(0.8)(hist, chain)
        rec["proc_chain"] = chain.numpy()
    np.savez_compressed(os.path.join(OUT, "gpt_infer.npz"), **rec)


TINY_DIFF = dict(model_channels=64, num_layers=2, in_channels=20, in_latent_channels=32, out_channels=40, dropout=0, num_heads=4,
                 layer_drop=0.0, unconditioned_percentage=0.0)


def gen_diffusion():
    """the reference's AA_diffusion (tiny config, deterministic weights) through
    SpacedDiffusion.training_losses with fixed t / noise (incl. t = 0: decoder-NLL branch), gradients, the surface of the full
    config, one AttentionBlock and one ResBlock in isolation, and three optimizer steps of the trainer's recipe."""
    from oracle import diffusion_ref as DR
    from ttts.diffusion.aa_model import AA_diffusion, DiffusionLayer, ResBlock
    from ttts.utils.utils import AttentionBlock
    from ttts.utils.diffusion import SpacedDiffusion, space_timesteps, get_named_beta_schedule
    import yaml
    full_cfg = yaml.safe_load(open("/root/reference/ttts/diffusion/config.yaml"))["aa_diffusion"]
    full = AA_diffusion(**full_cfg)
    rec = {"cfg": np.array(json.dumps(TINY_DIFF)), "full_cfg": np.array(json.dumps(full_cfg)),
           "full_surface": np.array(json.dumps([[k, list(v.shape)] for k, v in full.state_dict().items()]))}

    def fill(mod, gain=1.0):
        with torch.no_grad():
            for k, p in mod.named_parameters():
                p.copy_(DR.det_fill(k, p.shape, gain))
    g = torch.Generator().manual_seed(21)
    # ---- blocks
    ab = AttentionBlock(64, 4, relative_pos_embeddings=True); fill(ab)
    xa = torch.randn(2, 64, 37, generator=g).requires_grad_(True)
    ya = ab(xa); (ya * torch.linspace(-1, 1, 37)).sum().backward()
    rec.update({"ab_x": xa.detach().numpy(), "ab_y": ya.detach().numpy(), "ab_dx": xa.grad.numpy(),
                "ab_dtable": ab.relative_pos_embeddings.relative_attention_bias.weight.grad.numpy(),
                "ab_dqkv_w": ab.qkv.weight.grad.numpy()})
    rb = ResBlock(64, 64, 0, dims=1, use_scale_shift_norm=True); fill(rb)
    xr = torch.randn(2, 64, 29, generator=g).requires_grad_(True); er = torch.randn(2, 64, generator=g).requires_grad_(True)
    yr = rb(xr, er); (yr * torch.linspace(-1, 1, 29)).sum().backward()
    rec.update({"rb_x": xr.detach().numpy(), "rb_emb": er.detach().numpy(), "rb_y": yr.detach().numpy(), "rb_dx": xr.grad.numpy(),
                "rb_demb": er.grad.numpy(), "rb_dgamma_out": rb.out_layers[0].weight.grad.numpy()})
    # ---- model + loss
    m = AA_diffusion(**TINY_DIFF); fill(m, 0.7); m.train()
    d = SpacedDiffusion(use_timesteps=space_timesteps(1000, [1000]), model_mean_type="epsilon", model_var_type="learned_range",
                        loss_type="mse", betas=get_named_beta_schedule("linear", 1000), conditioning_free=False, conditioning_free_k=2.0)
    x0 = (torch.randn(3, 20, 48, generator=g) * 0.4).clamp(-1.2, 1.2); x0[0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995])
    lat = torch.randn(3, 32, 12, generator=g); ref = torch.randn(3, 20, 30, generator=g) * 0.4
    t = torch.tensor([0, 500, 999]); noise = torch.randn(3, 20, 48, generator=g)
    out = d.training_losses(model=m, x_start=x0, t=t, model_kwargs={"latent": lat, "refer": ref}, noise=noise)
    out["loss"].mean().backward()
    with torch.no_grad():
        x_t = d.q_sample(x0, t, noise=noise)
        model_out = m(x_t, t, latent=lat, refer=ref)
    rec.update({"x_start": x0.numpy(), "latent": lat.numpy(), "refer": ref.numpy(), "t": t.numpy(), "noise": noise.numpy(),
                "x_t": x_t.numpy(), "model_out": model_out.numpy(), "loss": out["loss"].detach().numpy(),
                "mse": out["mse"].detach().numpy(), "vb": out["vb"].detach().numpy()})
    names = [k for k, p in m.named_parameters()]
    rec["param_names"] = np.array(json.dumps(names))
    rec["grad_abs_sum"] = np.array([float(p.grad.abs().sum()) if p.grad is not None else -1.0 for _, p in m.named_parameters()])
    rec["grad_sum"] = np.array([float(p.grad.sum()) if p.grad is not None else 0.0 for _, p in m.named_parameters()])
    for k in ("layers.0.attn.qkv.weight", "time_embed.0.weight", "refer_enc.4.latents", "inp_block.weight",
              "layers.1.attn.relative_pos_embeddings.relative_attention_bias.weight", "out.2.weight",
              "conditioning_timestep_integrator.0.resblk.emb_layers.1.weight", "latent_conditioner.0.weight"):
        rec["grad:" + k] = dict(m.named_parameters())[k].grad.numpy()
    # diffusion tables
    for k in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_log_variance_clipped", "posterior_mean_coef1",
              "posterior_mean_coef2", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        rec["tab:" + k] = getattr(d, k)
    # host-side schedule helpers
    rec["space_50"] = np.array(sorted(space_timesteps(1000, [50])))
    rec["space_ddim25"] = np.array(sorted(space_timesteps(1000, "ddim25")))
    rec["space_10_15_20"] = np.array(sorted(space_timesteps(300, [10, 15, 20])))
    rec["betas_cosine_100"] = get_named_beta_schedule("cosine", 100)
    d50 = SpacedDiffusion(use_timesteps=space_timesteps(1000, [50]), model_mean_type="epsilon", model_var_type="learned_range",
                          loss_type="mse", betas=get_named_beta_schedule("linear", 1000))
    rec["spaced50_betas"], rec["spaced50_map"] = d50.betas, np.array(d50.timestep_map)
    rec["spaced50_post_logvar"] = d50.posterior_log_variance_clipped
    # unconditioned rows + dropped layer (the two random branches, forced)
    m.eval()
    with torch.no_grad():
        rec["model_out_eval"] = m(x_t, t, latent=lat, refer=ref).numpy()
        m.training = True; m.unconditioned_percentage = 2.0                 # rand < 2: every row unconditioned
        rec["model_out_uncond"] = m(x_t, t, latent=lat, refer=ref).numpy()
        m.unconditioned_percentage = 0.0
    # ---- three steps of the trainer recipe (train.py:119-120,194-199)
    m2 = AA_diffusion(**TINY_DIFF); fill(m2, 0.7); m2.train()
    opt = torch.optim.AdamW(m2.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: float(s / 1000) if s < 1000 else 1)
    sched.step()                                                           # step 0 has lr 0: start the record at step 1
    before = {k: p.detach().clone() for k, p in m2.named_parameters()}
    norms, losses = [], []
    for s_ in range(3):
        loss = d.training_losses(model=m2, x_start=x0, t=t, model_kwargs={"latent": lat, "refer": ref}, noise=noise)["loss"].mean()
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(m2.parameters(), 1.0)))
        opt.step(); opt.zero_grad(); sched.step()
        losses.append(float(loss))
    rec["step_norms"], rec["step_losses"] = np.array(norms), np.array(losses)
    rec["step_delta_abs"] = np.array([float((p.detach() - before[k]).abs().sum()) for k, p in m2.named_parameters()])
    np.savez_compressed(os.path.join(OUT, "diffusion.npz"), **rec)
    print("diffusion: loss", rec["loss"], "mse", rec["mse"], "vb", rec["vb"], "norms", norms, "unused params",
          int((rec["grad_abs_sum"] < 0).sum()), "tensors", len(names))


PEQ_CFGS = [dict(sampling_rate=32000, win_length=2048, hop_length=640, cutoff_lowpass=60, cutoff_highpass=10000, num_peak=8,
                 q_min=2, q_max=5, T=9000),
            dict(sampling_rate=22050, win_length=1024, hop_length=256, cutoff_lowpass=60, cutoff_highpass=10000, num_peak=8,
                 q_min=2, q_max=5, T=4096)]


def gen_peq():
    """the reference's Augment.forward (PEQ path, no Praat) and the three ParametricEqualizer responses
    on seeded clips, for vqvae/config.json's settings and a 22.05 kHz / 1024 / 256 set."""
    import types
    pm = types.ModuleType("parselmouth")      # imported (and used in an annotation) by augment/praat.py; Praat stage excluded
    pm.Sound = type("Sound", (), {})
    sys.modules.setdefault("parselmouth", pm)
    from ttts.vqvae.augment import Augment
    rec = {"cfgs": np.array(json.dumps(PEQ_CFGS))}
    for ci, c in enumerate(PEQ_CFGS):
        hp = types.SimpleNamespace(
            data=types.SimpleNamespace(sampling_rate=c["sampling_rate"], win_length=c["win_length"], hop_length=c["hop_length"]),
            train=types.SimpleNamespace(cutoff_lowpass=c["cutoff_lowpass"], cutoff_highpass=c["cutoff_highpass"],
                                        num_peak=c["num_peak"], q_min=c["q_min"], q_max=c["q_max"]))
        aug = Augment(hp)
        g = torch.Generator().manual_seed(77 + ci)
        B, T = 3, c["T"]
        t = torch.arange(T) / c["sampling_rate"]
        wav = sum(torch.rand(B, 1, generator=g) * torch.sin(2 * np.pi * (80. + 4000. * torch.rand(B, 1, generator=g)) * t[None]
                                                             + 6.28 * torch.rand(B, 1, generator=g)) for _ in range(6)) / 4.0
        wav = (wav + 0.05 * torch.randn(B, T, generator=g)).clamp(-1, 1).float()
        wav[2] *= 3.0                                                   # drives the clamp
        power = torch.rand(B, c["num_peak"] + 2, generator=g)
        gain = torch.rand(B, c["num_peak"] + 2, generator=g) * 24 - 12
        with torch.no_grad():
            out = aug(wav, quality_power=power, gain=gain)
            out_id = aug(wav)                                           # no equaliser: stft -> istft -> clamp -> normalise
            q = c["q_min"] * (c["q_max"] / c["q_min"]) ** power
            center = aug.peak_centers[None].repeat(B, 1)
            peaks = aug.peq.peaking_equalizer(center, gain[:, :-2], q[:, :-2])
            low = aug.peq.low_shelving(c["cutoff_lowpass"], gain[:, -2], q[:, -2])
            high = aug.peq.high_shelving(c["cutoff_highpass"], gain[:, -1], q[:, -1])
        k = "c%d_" % ci
        rec.update({k + "wav": wav.numpy(), k + "power": power.numpy(), k + "gain": gain.numpy(), k + "out": out.numpy(),
                    k + "out_identity": out_id.numpy(), k + "peaks": peaks.numpy(), k + "low": low.numpy(), k + "high": high.numpy(),
                    k + "peak_centers": aug.peak_centers.numpy()})
        print("peq cfg", ci, "out", tuple(out.shape), "peak |H|", float((torch.prod(peaks, 1) * low * high).abs().max()))
    np.savez_compressed(os.path.join(OUT, "vqvae_peq.npz"), **rec)


def gen_sampler():
    """Batches of the reference's DistributedBucketSampler (ttts/vqvae/dataset.py:212-307).  The module imports torchaudio /
    torchvision / pypinyin at import time (absent here), so only the class body is executed: its source segment is compiled
    from the reference file in this process; nothing of it is stored -- the fixture holds lengths, arguments and batches."""
    import ast
    src = open("/root/reference/ttts/vqvae/dataset.py").read()
    node = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "DistributedBucketSampler"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), "ref_sampler", "exec"), ns)
    Ref = ns["DistributedBucketSampler"]

    class DS:
        def __init__(self, lengths):
            self.lengths = lengths

        def __len__(self):
            return len(self.lengths)
    rng = np.random.default_rng(5)
    lengths = [int(v) for v in np.concatenate([rng.integers(20, 2100, 700), [32, 33, 300, 301, 1900, 1901]])]
    bounds = [32, 300, 400, 500, 600, 700, 800, 900, 1000, 1100, 1200, 1300, 1400, 1500, 1600, 1700, 1800, 1900]
    cases = []
    for world, bs, shuffle, bnd in ((1, 8, True, bounds), (2, 8, True, bounds), (4, 3, True, bounds), (2, 8, False, bounds),
                                   (2, 4, True, [32, 300, 305, 310, 2000])):
        per_rank = []
        for rank in range(world):
            smp = Ref(DS(lengths), bs, list(bnd), num_replicas=world, rank=rank, shuffle=shuffle)
            ep = {}
            for epoch in (1, 2):
                smp.set_epoch(epoch)
                ep[str(epoch)] = [list(map(int, b)) for b in smp]
            per_rank.append({"batches": ep, "len": len(smp), "boundaries_after": list(smp.boundaries),
                             "num_samples_per_bucket": list(smp.num_samples_per_bucket)})
        cases.append({"world": world, "batch_size": bs, "shuffle": shuffle, "boundaries": list(bnd), "ranks": per_rank})
    json.dump({"lengths": lengths, "cases": cases}, open(os.path.join(OUT, "sampler.json"), "w"))
    print("sampler:", [(c["world"], c["ranks"][0]["len"]) for c in cases])


if __name__ == "__main__":
    which = sys.argv[1:] or ["gpt", "vq", "mel", "vqvae", "disc", "flow", "attn", "step", "vqinfer", "peq", "infer", "diffusion", "sampler"]
    with torch.no_grad() if False else torch.enable_grad():
        if "gpt" in which:
            gen_gpt()
        if "vq" in which:
            gen_vq()
        if "mel" in which:
            gen_mel()
        if "vqvae" in which:
            gen_vqvae()
        if "disc" in which:
            gen_disc()
        if "flow" in which:
            gen_flow()
        if "attn" in which:
            gen_attn()
        if "step" in which:
            gen_step()
        if "vqinfer" in which:
            gen_vq_infer()
        if "peq" in which:
            gen_peq()
        if "infer" in which:
            gen_infer()
        if "diffusion" in which:
            gen_diffusion()
        if "sampler" in which:
            gen_sampler()
    print("fixtures:", {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})
