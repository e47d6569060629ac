"""CPU: the inference restatement in oracle/gpt_ref.py (latent export, cache-less decoding, logits processors) against the
reference-generated fixture tests/golden/gpt_infer.npz (tools/make_goldens.py infer)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gpt_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "gpt_infer.npz"))


@pytest.fixture(scope="module")
def tiny(gold):
    cfg = json.loads(str(gold["cfg_json"]))
    return cfg, R.det_state_dict(cfg)


def test_latent_export(gold, tiny):
    cfg, sd = tiny
    g1 = np.load(os.path.join(GOLD, "gpt_tiny.npz"))
    T = lambda k: torch.from_numpy(g1[k])
    lat = R.latent_forward(sd, cfg, T("text"), T("text_lengths"), T("mel"), T("wav_lengths"))
    assert lat.shape == gold["latent"].shape == (2, 24, 64)            # mel_len tokens: the forward's +2 stripped again
    np.testing.assert_allclose(lat.numpy(), gold["latent"], atol=2e-5)


def test_greedy_decode_tokens_and_logits(gold, tiny):
    cfg, sd = tiny
    codes, raw = R.generate(sd, cfg, torch.from_numpy(gold["itext"]), torch.from_numpy(gold["prompt"]), 10)
    np.testing.assert_allclose(torch.stack(raw).numpy(), gold["greedy_logits"], atol=2e-5)
    assert np.array_equal(codes.numpy(), gold["greedy_codes"])          # fp32 both sides; smallest top-2 margin 3e-4
    text_inp, mel = R.inference_inputs(cfg, torch.from_numpy(gold["itext"]), torch.from_numpy(gold["prompt"]))
    np.testing.assert_allclose(R.inference_logits(sd, cfg, text_inp, mel).numpy(), gold["first_pass_logits"], atol=2e-5)


def test_logits_processors(gold):
    s, h = torch.from_numpy(gold["proc_scores"]), torch.from_numpy(gold["proc_hist"])
    eq = lambda a, k: np.testing.assert_array_equal(a.numpy(), gold[k])
    eq(R.repetition_penalty_(s.clone(), h, 2.0), "proc_rep2")
    eq(s / 0.8, "proc_temp08")
    eq(R.top_k_filter(s.clone(), 50), "proc_topk50")
    eq(R.top_p_filter(s.clone(), 0.8), "proc_topp08")
    eq(R.typical_filter(s.clone(), 0.9), "proc_typical09")
    eq(R.process_logits(s.clone(), h, repetition_penalty=2.0, temperature=0.8, top_k=50, top_p=0.8), "proc_chain")
    assert int(np.isfinite(gold["proc_chain"]).sum(-1).min()) >= 1


def test_eos_pads_finished_rows(tiny):
    """A row that emits stop_mel_token keeps emitting it (pad = eos) while the others continue; all rows done -> stop."""
    cfg, sd = tiny
    c = R.full_cfg(cfg)
    calls = []

    def choose(scores):
        calls.append(1)
        n = len(calls)
        return torch.tensor([c["stop_mel_token"] if n >= 2 else 5, c["stop_mel_token"] if n >= 4 else 7])
    codes, _ = R.generate(sd, cfg, torch.randint(1, 255, (2, 5)), torch.randint(0, 1024, (2, 3)), 10, choose=choose)
    stop = c["stop_mel_token"]
    assert codes.tolist() == [[5, stop, stop, stop], [7, 7, 7, stop]]
