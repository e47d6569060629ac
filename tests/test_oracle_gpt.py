"""The oracle (CPU restatement) against the fixtures generated from the imported reference.
Pins oracle/gpt_ref.py to ttts/gpt/model.py + HF GPT-2 (+ torch AdamW / LambdaLR / clip)."""
import json
import os

import numpy as np
import torch

from oracle import gpt_ref


def _sample(t, n=4096):
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)].numpy()


def test_surface_matches_reference(golden_dir):
    surf = json.load(open(os.path.join(golden_dir, "surface.json")))
    spec = gpt_ref.state_dict_spec(surf["gpt_config"])
    assert [[k, list(s)] for k, s in spec] == [[k, s] for k, s, _ in surf["gpt"]]
    assert all(dt == "torch.float32" for _, _, dt in surf["gpt"])
    assert len(spec) == 84
    assert sum(int(np.prod(s)) for _, s in spec) == 21462275


def test_tiny_forward_backward(golden_dir):
    g = np.load(os.path.join(golden_dir, "gpt_tiny.npz"))
    cfg = json.loads(str(g["cfg_json"]))
    sd = {k: v.requires_grad_(True) for k, v in gpt_ref.det_state_dict(cfg).items()}
    args = [torch.from_numpy(g[k]) for k in ("text", "text_lengths", "mel", "wav_lengths")]
    lt, lm, logits = gpt_ref.unified_voice_forward(sd, cfg, *args)
    np.testing.assert_allclose(lt.item(), g["loss_text"], rtol=2e-6)
    np.testing.assert_allclose(lm.item(), g["loss_mel"], rtol=2e-6)
    np.testing.assert_allclose(logits.detach().numpy(), g["mel_logits"], rtol=1e-4, atol=2e-6)
    (lt * 0.01 + lm).backward()
    gn = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values())))
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=1e-5)
    for k, v in sd.items():
        ref = g["grad:" + k]
        np.testing.assert_allclose(_sample(v.grad), ref, rtol=2e-4, atol=1e-7 + 1e-5 * np.abs(ref).max(), err_msg=k)


def test_set_mel_padding_does_not_mutate_and_matches(golden_dir):
    g = np.load(os.path.join(golden_dir, "gpt_tiny.npz"))
    cfg = json.loads(str(g["cfg_json"]))
    mel = torch.from_numpy(g["mel"])
    before = mel.clone()
    _, _, mel_inp, mel_tar = gpt_ref.prepare_tokens(torch.from_numpy(g["text"]), torch.from_numpy(g["text_lengths"]),
                                                    mel, torch.from_numpy(g["wav_lengths"]), cfg)
    assert torch.equal(mel, before)
    assert mel_inp.shape == (2, 26) and int(mel_inp[0, 0]) == 1024 and int(mel_tar[0, -1]) == 1025
    assert torch.all(mel_tar[1, 18:] == 1025)  # wav 17*1024+5 -> 17 codes + 1, rest STOP


def test_full_config_b1(golden_dir):
    g = np.load(os.path.join(golden_dir, "gpt_full_b1.npz"))
    torch.set_num_threads(8)
    sd = {k: v.requires_grad_(True) for k, v in gpt_ref.det_state_dict(None).items()}
    batch = gpt_ref.synthetic_batch(B=1, seed=int(g["seed"]))
    lt, lm, logits = gpt_ref.unified_voice_forward(sd, None, *batch)
    np.testing.assert_allclose(lt.item(), g["loss_text"], rtol=5e-6)
    np.testing.assert_allclose(lm.item(), g["loss_mel"], rtol=5e-6)
    np.testing.assert_allclose(logits.detach()[0, ::64, ::64].numpy(), g["logits_slice"], rtol=2e-4, atol=5e-6)
    (lt * 0.01 + lm).backward()
    gn = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values())))
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-5)
    for key in g.files:
        if key.startswith("grad:"):
            ref = g[key]
            np.testing.assert_allclose(_sample(sd[key[5:]].grad, 2048), ref, rtol=1e-3,
                                       atol=1e-7 + 2e-5 * np.abs(ref).max(), err_msg=key)


def test_optimizer_steps(golden_dir):
    g = np.load(os.path.join(golden_dir, "gpt_step.npz"))
    gt = np.load(os.path.join(golden_dir, "gpt_tiny.npz"))
    cfg = json.loads(str(g["cfg_json"]))
    sd0 = gpt_ref.det_state_dict(cfg)
    sd = {k: v.clone() for k, v in sd0.items()}
    opt = gpt_ref.new_opt_state(sd)
    batch = [torch.from_numpy(gt[k]) for k in ("text", "text_lengths", "mel", "wav_lengths")]
    for s in range(len(g["losses"])):
        out = gpt_ref.gpt_train_step(sd, opt, batch, cfg)
        np.testing.assert_allclose(out["loss"], g["losses"][s], rtol=2e-6)
        np.testing.assert_allclose(out["grad_norm"], g["grad_norms"][s], rtol=1e-5)
    for k in sd:
        np.testing.assert_allclose(_sample(opt["m"][k], 1024), g["m:" + k], rtol=1e-3, atol=1e-9, err_msg=k)
        np.testing.assert_allclose(_sample(opt["v"][k], 1024), g["v:" + k], rtol=1e-3, atol=1e-12, err_msg=k)
        d = _sample(sd[k].double() - sd0[k].double(), 1024)
        ref = g["delta:" + k]
        np.testing.assert_allclose(d, ref, rtol=2e-2, atol=2e-7 * max(1.0, float(np.abs(_sample(sd0[k], 1024)).max())),
                                   err_msg=k)


def test_bf16_mode_is_close_to_fp32(golden_dir):
    g = np.load(os.path.join(golden_dir, "gpt_tiny.npz"))
    cfg = json.loads(str(g["cfg_json"]))
    sd = gpt_ref.det_state_dict(cfg)
    args = [torch.from_numpy(g[k]) for k in ("text", "text_lengths", "mel", "wav_lengths")]
    lt, lm, _ = gpt_ref.unified_voice_forward(sd, cfg, *args, bf16=True)
    np.testing.assert_allclose(lt.item(), g["loss_text"], rtol=1e-2)
    np.testing.assert_allclose(lm.item(), g["loss_mel"], rtol=1e-2)
