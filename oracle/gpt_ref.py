"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference GPT train path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module,
and only as the checker / reported CPU baseline.  The product (`ttts_amd/`) never imports it.

What is restated (reference = /root/reference, adelacvg/ttts; HF = transformers 5.15.0, the GPT-2
implementation the reference instantiates, an unpinned third-party dependency -- SURVEY.md F4/F5):

* `UnifiedVoice.forward`                        ttts/gpt/model.py:453-510
* `set_mel_padding`                             ttts/gpt/model.py:402-414
* `build_aligned_inputs_and_targets`            ttts/gpt/model.py:397-400
* `LearnedPositionEmbeddings.forward`           ttts/gpt/model.py:237-239
* `get_logits` (text first)                     ttts/gpt/model.py:416-443
* HF `GPT2Block` / `GPT2Attention` / `GPT2MLP`  transformers/models/gpt2/modeling_gpt2.py:53-72,144-309
  (eager attention form: scale dh^-1/2, causal mask, fp32 softmax; `gelu_new` tanh form)
* `Trainer.train` step body                     ttts/gpt/train.py:96-121 (loss weights, grad-norm,
  clip 1.0, AdamW(lr, betas 0.9/0.96, wd 0.01), LambdaLR warm-up :36-40,57)

Parity pin: this restatement is checked against fixtures produced by importing the reference itself
(`tools/make_goldens.py` -> `tests/golden/gpt_*.npz`); the reference has no tests of its own for this
path, so the fixtures pin transformers-5.15 / torch-2.10 behaviour as observed.

The state dict uses the reference's keys verbatim (84 tensors, HF `Conv1D` weights stored [in, out]).
Plain torch ops on whatever device the tensors live on; fp32 by default.  `bf16=True` inserts the
rounding points CUDA/HIP autocast(bfloat16) places in the reference (matmul operands and outputs in
bf16, LayerNorm / softmax / cross-entropy in fp32, fp32 residual stream).
"""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

GPT_CONFIG = {  # ttts/gpt/config.json:16-29
    "model_dim": 512, "max_mel_tokens": 1600, "max_text_tokens": 800, "heads": 8,
    "use_mel_codes_as_input": True, "layers": 6, "number_text_tokens": 256,
    "number_mel_codes": 1026, "start_mel_token": 1024, "stop_mel_token": 1025,
    "start_text_token": 255, "train_solo_embeddings": False,
}
TRAIN_CONFIG = {"lr": 1e-4, "text_weight": 0.01, "mel_weight": 1, "accumulate_num": 1}  # config.json:2-12


def full_cfg(cfg=None):
    """Fill UnifiedVoice constructor defaults (ttts/gpt/model.py:293-297)."""
    c = dict(layers=8, model_dim=512, heads=8, max_text_tokens=120, max_mel_tokens=250,
             mel_length_compression=1024, number_text_tokens=256, start_text_token=None,
             number_mel_codes=8194, start_mel_token=8192, stop_mel_token=8193, types=1)
    c.update(cfg or GPT_CONFIG)
    if c["start_text_token"] is None:
        c["start_text_token"] = c["number_text_tokens"] * c["types"]
    c["stop_text_token"] = 0
    return c


def state_dict_spec(cfg):
    """(key, shape) list in the reference's state-dict order (golden G0)."""
    c = full_cfg(cfg)
    d, L = c["model_dim"], c["layers"]
    spec = [("text_embedding.weight", (c["number_text_tokens"] * c["types"] + 1, d)),
            ("mel_embedding.weight", (c["number_mel_codes"], d))]
    for i in range(L):
        p = f"gpt.h.{i}."
        spec += [(p + "ln_1.weight", (d,)), (p + "ln_1.bias", (d,)),
                 (p + "attn.c_attn.weight", (d, 3 * d)), (p + "attn.c_attn.bias", (3 * d,)),
                 (p + "attn.c_proj.weight", (d, d)), (p + "attn.c_proj.bias", (d,)),
                 (p + "ln_2.weight", (d,)), (p + "ln_2.bias", (d,)),
                 (p + "mlp.c_fc.weight", (d, 4 * d)), (p + "mlp.c_fc.bias", (4 * d,)),
                 (p + "mlp.c_proj.weight", (4 * d, d)), (p + "mlp.c_proj.bias", (d,))]
    spec += [("gpt.ln_f.weight", (d,)), ("gpt.ln_f.bias", (d,)),
             ("mel_pos_embedding.emb.weight", (c["max_mel_tokens"] + 2, d)),
             ("text_pos_embedding.emb.weight", (c["max_text_tokens"] + 2, d)),
             ("final_norm.weight", (d,)), ("final_norm.bias", (d,)),
             ("text_head.weight", (c["number_text_tokens"] * c["types"] + 1, d)),
             ("text_head.bias", (c["number_text_tokens"] * c["types"] + 1,)),
             ("mel_head.weight", (c["number_mel_codes"], d)), ("mel_head.bias", (c["number_mel_codes"],))]
    return spec


def det_fill(name, shape, scale=0.02):
    """Deterministic, construction-order-independent parameter fill shared by the golden generator,
    the oracle tests and the GPU parity tests (so 21 M-parameter fixtures need not be stored).
    LayerNorm gains are centred on 1, everything else on 0."""
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    a = rng.standard_normal(size=shape, dtype=np.float32) * np.float32(scale)
    is_ln_gain = name.endswith("weight") and len(shape) == 1
    if is_ln_gain:
        a = a + np.float32(1.0)
    return torch.from_numpy(a)


def det_state_dict(cfg, scale=0.02):
    return {k: det_fill(k, s, scale) for k, s in state_dict_spec(cfg)}


def _r(x, bf16):
    """bf16 rounding point (autocast) -- identity in fp32 mode."""
    return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


def gelu_new(x):
    # transformers activations.NewGELUActivation (tanh form)
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def _linear_conv1d(x, w, b, bf16):
    """HF Conv1D: y = x @ W[in,out] + b (modeling_utils Conv1D.forward, addmm)."""
    y = _r(x, bf16) @ _r(w, bf16) + (_r(b, bf16) if bf16 else b)
    return _r(y, bf16)


def gpt2_attention(h, sd, pfx, heads, bf16, dropout_p=0.0):
    """modeling_gpt2.py:144-226 with eager_attention_forward :53-72 (causal)."""
    B, S, D = h.shape
    dh = D // heads
    qkv = _linear_conv1d(h, sd[pfx + "c_attn.weight"], sd[pfx + "c_attn.bias"], bf16)
    q, k, v = qkv.split(D, dim=2)
    q = q.view(B, S, heads, dh).transpose(1, 2)
    k = k.view(B, S, heads, dh).transpose(1, 2)
    v = v.view(B, S, heads, dh).transpose(1, 2)
    att = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    causal = torch.ones(S, S, dtype=torch.bool, device=h.device).tril()
    att = att.masked_fill(~causal, float("-inf"))
    att = torch.softmax(att.float(), dim=-1)
    att = _r(att, bf16)
    if dropout_p > 0:
        att = F.dropout(att, dropout_p, True)
    o = _r(att @ v, bf16)
    o = o.transpose(1, 2).reshape(B, S, D)
    o = _linear_conv1d(o, sd[pfx + "c_proj.weight"], sd[pfx + "c_proj.bias"], bf16)
    if dropout_p > 0:
        o = F.dropout(o, dropout_p, True)
    return o


def gpt2_block(x, sd, i, heads, bf16, dropout_p=0.0):
    """modeling_gpt2.py:246-309 (pre-LN residual wiring; LayerNorm eps 1e-5 in fp32)."""
    p = f"gpt.h.{i}."
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
    x = gpt2_attention(h, sd, p + "attn.", heads, bf16, dropout_p) + x
    h = F.layer_norm(x, (D,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
    h = _linear_conv1d(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"], bf16)
    h = _r(gelu_new(h), bf16)
    h = _linear_conv1d(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"], bf16)
    if dropout_p > 0:
        h = F.dropout(h, dropout_p, True)
    return x + h


def set_mel_padding(mel, wav_lengths, c):
    """ttts/gpt/model.py:402-414 (the reference mutates its argument; this restatement clones)."""
    mel = mel.clone()
    mel_lengths = torch.div(wav_lengths, c["mel_length_compression"], rounding_mode="trunc")
    for b in range(len(mel_lengths)):
        end = int(mel_lengths[b]) + 1
        if end < mel.shape[-1]:
            mel[b, end:] = c["stop_mel_token"]
    return mel


def prepare_tokens(text_inputs, text_lengths, mel_codes, wav_lengths, cfg, clip_inputs=True):
    """Token plumbing of UnifiedVoice.forward (model.py:474-489): clip, stop/start padding.
    Returns (text_inp, text_tar, mel_inp, mel_tar) int64."""
    c = full_cfg(cfg)
    if clip_inputs:
        text_inputs = text_inputs[:, :int(text_lengths.max())]
        mel_codes = mel_codes[:, :int(wav_lengths.max()) // c["mel_length_compression"]]
    mel_codes = set_mel_padding(mel_codes, wav_lengths, c)
    text_inputs = F.pad(text_inputs, (0, 1), value=c["stop_text_token"])
    mel_codes = F.pad(mel_codes, (0, 1), value=c["stop_mel_token"])
    text_inp = F.pad(text_inputs, (1, 0), value=c["start_text_token"])
    text_tar = F.pad(text_inputs, (0, 1), value=c["stop_text_token"])
    mel_inp = F.pad(mel_codes, (1, 0), value=c["start_mel_token"])
    mel_tar = F.pad(mel_codes, (0, 1), value=c["stop_mel_token"])
    return text_inp, text_tar, mel_inp, mel_tar


def unified_voice_forward(sd, cfg, text_inputs, text_lengths, mel_codes, wav_lengths,
                          bf16=False, dropout_p=0.0, clip_inputs=True, return_hidden=False):
    """UnifiedVoice.forward (text_first=True, raw_mels=None) -> (loss_text, loss_mel, mel_logits).
    mel_logits has the reference's permuted layout (B, classes, positions)."""
    c = full_cfg(cfg)
    D, heads = c["model_dim"], c["heads"]
    text_inp, text_tar, mel_inp, mel_tar = prepare_tokens(text_inputs, text_lengths, mel_codes, wav_lengths,
                                                          cfg, clip_inputs)
    Tt, Tm = text_inp.shape[1], mel_inp.shape[1]
    text_emb = F.embedding(text_inp, sd["text_embedding.weight"]) + sd["text_pos_embedding.emb.weight"][:Tt]
    mel_emb = F.embedding(mel_inp, sd["mel_embedding.weight"]) + sd["mel_pos_embedding.emb.weight"][:Tm]
    x = torch.cat([text_emb, mel_emb], dim=1)  # wpe is the null embedding (model.py:259-260)
    if dropout_p > 0:
        x = F.dropout(x, dropout_p, True)  # GPT2Model.drop (embd_pdrop)
    for i in range(c["layers"]):
        x = gpt2_block(x, sd, i, heads, bf16, dropout_p)
    x = F.layer_norm(x, (D,), sd["gpt.ln_f.weight"], sd["gpt.ln_f.bias"], 1e-5)
    enc = F.layer_norm(x, (D,), sd["final_norm.weight"], sd["final_norm.bias"], 1e-5)
    if return_hidden:
        return enc
    tl = _r(_r(enc[:, :Tt], bf16) @ _r(sd["text_head.weight"], bf16).t() + _r(sd["text_head.bias"], bf16), bf16)
    ml = _r(_r(enc[:, -Tm:], bf16) @ _r(sd["mel_head.weight"], bf16).t() + _r(sd["mel_head.bias"], bf16), bf16)
    text_logits, mel_logits = tl.permute(0, 2, 1), ml.permute(0, 2, 1)
    loss_text = F.cross_entropy(text_logits.float(), text_tar)
    loss_mel = F.cross_entropy(mel_logits.float(), mel_tar)
    return loss_text, loss_mel, mel_logits


# ---------------------------------------------------------------------------------------------------------
# train step (ttts/gpt/train.py:96-121)

def warmup(step):
    """ttts/gpt/train.py:36-40"""
    return float(step / 500) if step < 500 else 1


def new_opt_state(sd):
    return {"step": 0, "m": {k: torch.zeros_like(v) for k, v in sd.items()},
            "v": {k: torch.zeros_like(v) for k, v in sd.items()}}


def clip_and_adamw_(sd, grads, opt, base_lr=1e-4, betas=(0.9, 0.96), eps=1e-8, wd=0.01, max_norm=1.0):
    """get_grad_norm (train.py:22-31) + clip_grad_norm_(…, 1.0) (:115) + AdamW.step (:56,118)
    + LambdaLR(warmup) (:57,120).  `opt['step']` counts completed optimizer steps; LambdaLR's
    constructor already applied warmup(0) = 0, so the first step runs with lr = 0."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    lr = base_lr * warmup(opt["step"])
    t = opt["step"] + 1
    b1, b2 = betas
    for k, p in sd.items():
        g = grads[k] * coef
        p.mul_(1.0 - lr * wd)
        m, v = opt["m"][k], opt["v"][k]
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v.sqrt() / math.sqrt(1.0 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1.0 - b1 ** t))
    opt["step"] = t
    return float(total)


def gpt_train_step(sd, opt, batch, cfg=None, train=None, bf16=False, dropout_p=0.0):
    """One optimizer step.  `sd` tensors are leaf fp32 tensors updated in place.
    batch = (padded_text, text_lengths, padded_qmel, wav_lens) (gpt/dataset.py:91-97, train.py:104-105)."""
    tr = dict(TRAIN_CONFIG)
    tr.update(train or {})
    leaves = {k: v.detach().requires_grad_(True) for k, v in sd.items()}
    lt, lm, _ = unified_voice_forward(leaves, cfg, *batch, bf16=bf16, dropout_p=dropout_p)
    loss = (lt * tr["text_weight"] + lm * tr["mel_weight"]) / tr["accumulate_num"]
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    with torch.no_grad():
        gn = clip_and_adamw_(sd, grads, opt, base_lr=tr["lr"])
    return {"loss": float(loss.detach()), "loss_text": float(lt.detach()), "loss_mel": float(lm.detach()), "grad_norm": gn, "grads": grads}


def synthetic_batch(B=8, text_len=128, mel_len=1024, seed=1234, cfg=None):
    """text int64 (B,128) in [1,255), mel int64 (B,1024) in [0,1024),
    wav_lengths = 1024*1024 (nothing clipped), seeded on CPU."""
    c = full_cfg(cfg)
    g = torch.Generator().manual_seed(seed)
    text = torch.randint(1, 255, (B, text_len), generator=g, dtype=torch.int64)
    mel = torch.randint(0, c["start_mel_token"], (B, mel_len), generator=g, dtype=torch.int64)
    tl = torch.full((B,), text_len, dtype=torch.int64)
    wl = torch.full((B,), mel_len * c["mel_length_compression"], dtype=torch.int64)
    return text, tl, mel, wl


# ---------------------------------------------------------------------------------------------------------
# inference: latent export and autoregressive decoding

def latent_forward(sd, cfg, text_inputs, text_lengths, mel_codes, wav_lengths, bf16=False, clip_inputs=False):
    """UnifiedVoice.forward(..., return_latent=True) (ttts/gpt/model.py:429-430,499-501): final_norm hidden states of the
    mel positions without the two tokens the forward added -> (B, mel_len, D)."""
    c = full_cfg(cfg)
    _, _, mel_inp, _ = prepare_tokens(text_inputs, text_lengths, mel_codes, wav_lengths, cfg, clip_inputs)
    enc = unified_voice_forward(sd, cfg, text_inputs, text_lengths, mel_codes, wav_lengths, bf16=bf16,
                                clip_inputs=clip_inputs, return_hidden=True)
    return enc[:, -mel_inp.shape[1]:][:, :-2]


def inference_logits(sd, cfg, text_inp, mel_tokens, bf16=False):
    """One full (cache-less) pass of GPT2InferenceModel.forward (ttts/gpt/model.py:109-184) as inference_speech drives it
    (:533-562, post_init_gpt2_config(kv_cache=False) as in api_zh.py:52): the cached prefix is the text embedding,
    `mel_tokens` (B, n) = [start_mel, prompt codes..., generated...] get mel_embedding + positions 0..n-1, the stack runs
    causally over [text ; mel], lm_head = final_norm -> mel_head.  Returns fp32 logits (B, Tt + n, classes)."""
    c = full_cfg(cfg)
    D, heads = c["model_dim"], c["heads"]
    Tt, n = text_inp.shape[1], mel_tokens.shape[1]
    text_emb = F.embedding(text_inp, sd["text_embedding.weight"]) + sd["text_pos_embedding.emb.weight"][:Tt]
    mel_emb = F.embedding(mel_tokens, sd["mel_embedding.weight"]) + sd["mel_pos_embedding.emb.weight"][:n]
    x = torch.cat([text_emb, mel_emb], dim=1)
    for i in range(c["layers"]):
        x = gpt2_block(x, sd, i, heads, bf16)
    x = F.layer_norm(x, (D,), sd["gpt.ln_f.weight"], sd["gpt.ln_f.bias"], 1e-5)
    enc = F.layer_norm(x, (D,), sd["final_norm.weight"], sd["final_norm.bias"], 1e-5)
    return (_r(enc, bf16) @ _r(sd["mel_head.weight"], bf16).t() + sd["mel_head.bias"]).float()


def inference_inputs(cfg, text_inputs, mel_codes):
    """inference_speech's input assembly (model.py:536-547): text -> [start, text, stop]; prompt -> [start_mel, codes]."""
    c = full_cfg(cfg)
    text = F.pad(text_inputs, (0, 1), value=c["stop_text_token"])
    text_inp = F.pad(text, (1, 0), value=c["start_text_token"])
    mel_inp = F.pad(mel_codes, (1, 0), value=c["start_mel_token"])
    return text_inp, mel_inp


# transformers.generation.logits_process (third-party, unpinned; 5.15.0 installed here) + the reference's own
# TypicalLogitsWarper (ttts/utils/typical_sampling.py:5-35); order of application as GenerationMixin._get_logits_processor
# builds it for inference_speech: repetition penalty -> typical (custom list) -> temperature -> top-k -> top-p.

def repetition_penalty_(scores, input_ids, penalty):
    score = torch.gather(scores, 1, input_ids)
    score = torch.where(score < 0, score * penalty, score / penalty)
    return scores.scatter(1, input_ids, score)


def typical_filter(scores, mass=0.9):
    normalized = torch.log_softmax(scores, dim=-1)
    p = torch.exp(normalized)
    ent = -(normalized * p).nansum(-1, keepdim=True)
    shifted = torch.abs((-normalized) - ent)
    sorted_scores, sorted_idx = torch.sort(shifted, descending=False)
    sorted_logits = scores.gather(-1, sorted_idx)
    cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    last_ind = (cum < mass).sum(dim=1)
    last_ind = last_ind.clamp(min=0, max=scores.shape[-1] - 1)
    remove_sorted = sorted_scores > sorted_scores.gather(1, last_ind.view(-1, 1))
    remove = remove_sorted.scatter(1, sorted_idx, remove_sorted)
    return scores.masked_fill(remove, float("-inf"))


def top_k_filter(scores, top_k):
    top_k = min(top_k, scores.shape[-1])
    kth = torch.topk(scores, top_k)[0][..., -1, None]
    return scores.masked_fill(scores < kth, float("-inf"))


def top_p_filter(scores, top_p):
    sorted_logits, sorted_idx = torch.sort(scores, descending=False)
    cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    remove_sorted = cum <= (1 - top_p)
    remove_sorted[..., -1:] = False
    remove = remove_sorted.scatter(1, sorted_idx, remove_sorted)
    return scores.masked_fill(remove, float("-inf"))


def process_logits(scores, input_ids, repetition_penalty=None, typical_mass=None, temperature=None, top_k=None, top_p=None):
    if repetition_penalty is not None and repetition_penalty != 1.0:
        scores = repetition_penalty_(scores, input_ids, repetition_penalty)
    if typical_mass is not None:
        scores = typical_filter(scores, typical_mass)
    if temperature is not None and temperature != 1.0:
        scores = scores / temperature
    if top_k is not None and top_k > 0:
        scores = top_k_filter(scores, top_k)
    if top_p is not None and top_p < 1.0:
        scores = top_p_filter(scores, top_p)
    return scores


def generate(sd, cfg, text_inputs, mel_codes, max_generate_length, bf16=False, choose=None, **proc):
    """inference_speech (model.py:533-562) with the sample loop of GenerationMixin._sample: full recompute per token,
    next = choose(processed scores) (default: argmax = do_sample False), finished rows emit pad (= stop_mel_token),
    stops at max_length = prompt + max_generate_length or when every row has produced stop_mel_token.
    Returns (codes (B, <= max_generate_length), per-step raw logits list)."""
    c = full_cfg(cfg)
    text_inp, mel = inference_inputs(cfg, text_inputs, mel_codes)
    Tt, trunc = text_inp.shape[1], text_inp.shape[1] + mel.shape[1]
    unfinished = torch.ones(mel.shape[0], dtype=torch.bool)
    raw = []
    for _ in range(max_generate_length):
        logits = inference_logits(sd, cfg, text_inp, mel, bf16)[:, -1]
        raw.append(logits)
        # HF hands the processors the whole input_ids row: the fake text slots (value 1) and every mel token so far
        ids = torch.cat([torch.ones(mel.shape[0], Tt, dtype=torch.long), mel], dim=1)
        scores = process_logits(logits.clone(), ids, **proc)
        nxt = scores.argmax(-1) if choose is None else choose(scores)
        nxt = torch.where(unfinished, nxt, torch.full_like(nxt, c["stop_mel_token"]))
        mel = torch.cat([mel, nxt[:, None]], dim=1)
        unfinished = unfinished & (nxt != c["stop_mel_token"])
        if not bool(unfinished.any()):
            break
    return mel[:, trunc - Tt:], raw
