"""`UnifiedVoice` with the reference's constructor / forward signature and state-dict surface
(ttts/gpt/model.py:292-510; 84 tensors, HF Conv1D weights stored [in, out]), executing on `GptEngine`.

Drop-in scope: the live TRAINING path -- `text_first=True`, `raw_mels=None`, `use_mel_codes_as_input=True`
(ttts/gpt/config.json).  Other branches of the reference forward (attention maps, raw-mel encoder, solo
embeddings, inference wrappers) raise NotImplementedError instead of silently taking an eager path.

Parameters are `nn.Parameter` VIEWS into the engine's flat fp32 arena and their `.grad` are views into the flat
gradient arena: `loss.backward()` launches the fused HIP backward, which accumulates straight into those views
(so `p.grad` is always current; use `FusedAdamW` or `GptEngine.optimizer_step` -- a stock torch optimizer with
`zero_grad(set_to_none=True)` would detach the views).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .decode import GptDecoder
from .engine import GptEngine, resolve_config


def prepare_tokens(c, text_inputs, text_lengths, mel_codes, wav_lengths, clip_inputs=True):
    """Token plumbing of UnifiedVoice.forward (ttts/gpt/model.py:474-489,397-414): clip to the batch maximum,
    rewrite mel padding to STOP, append STOP, build (input, target) pairs with START / STOP.
    Works on CPU or GPU tensors; the two `.max()` reads are host syncs only if the lengths live on the GPU
    (the reference does the same reads, model.py:477,479).  Does NOT mutate its arguments."""
    if clip_inputs:
        text_inputs = text_inputs[:, :int(text_lengths.max())]
        mel_codes = mel_codes[:, :int(wav_lengths.max()) // c["mel_length_compression"]]
    # set_mel_padding, vectorised: positions >= wav_len // compression + 1 become STOP
    mel_lengths = torch.div(wav_lengths, c["mel_length_compression"], rounding_mode="trunc") + 1
    pos = torch.arange(mel_codes.shape[-1], device=mel_codes.device)[None, :]
    mel_codes = torch.where(pos >= mel_lengths[:, None].to(mel_codes.device), c["stop_mel_token"], mel_codes)
    text_inputs = F.pad(text_inputs, (0, 1), value=c["stop_text_token"])
    mel_codes = F.pad(mel_codes, (0, 1), value=c["stop_mel_token"])
    text_inp = F.pad(text_inputs, (1, 0), value=c["start_text_token"])
    text_tar = F.pad(text_inputs, (0, 1), value=c["stop_text_token"])
    mel_inp = F.pad(mel_codes, (1, 0), value=c["start_mel_token"])
    mel_tar = F.pad(mel_codes, (0, 1), value=c["stop_mel_token"])
    return text_inp.long(), text_tar.long(), mel_inp.long(), mel_tar.long()


class _Node(nn.Module):
    """Plain container so parameters get the reference's dotted names."""


class _GptStep(torch.autograd.Function):
    """forward = fused HIP forward; backward = fused HIP backward accumulating into the gradient arena."""

    @staticmethod
    def forward(ctx, module, anchor):
        eng = module.engine
        eng.training = module.training
        eng.forward()
        ctx.module = module
        ctx.set_materialize_grads(False)
        losses = eng.b["losses"]
        return losses[0].clone(), losses[1].clone(), eng.mel_logits()

    @staticmethod
    def backward(ctx, g_text, g_mel, g_logits):
        if g_logits is not None:
            raise NotImplementedError("gradients through mel_logits are not part of the training path")
        eng = ctx.module.engine
        dev = eng.device
        one = None
        if g_text is None or g_mel is None:
            one = torch.zeros((), dtype=torch.float32, device=dev)
        gt = (g_text if g_text is not None else one).to(torch.float32).contiguous()
        gm = (g_mel if g_mel is not None else one).to(torch.float32).contiguous()
        eng.backward(1.0, 1.0, gt, gm)
        ctx.module.attach_grads()
        return None, None


class UnifiedVoice(nn.Module):
    def __init__(self, layers=8, model_dim=512, heads=8, max_text_tokens=120, max_mel_tokens=250,
                 max_conditioning_inputs=1, mel_length_compression=1024, number_text_tokens=256,
                 start_text_token=None, number_mel_codes=8194, start_mel_token=8192, stop_mel_token=8193,
                 train_solo_embeddings=False, use_mel_codes_as_input=True, checkpointing=True, types=1,
                 device=None, dropout_p=0.1, seed=0):
        super().__init__()
        cfg = dict(layers=layers, model_dim=model_dim, heads=heads, max_text_tokens=max_text_tokens,
                   max_mel_tokens=max_mel_tokens, max_conditioning_inputs=max_conditioning_inputs,
                   mel_length_compression=mel_length_compression, number_text_tokens=number_text_tokens,
                   start_text_token=start_text_token, number_mel_codes=number_mel_codes,
                   start_mel_token=start_mel_token, stop_mel_token=stop_mel_token,
                   train_solo_embeddings=train_solo_embeddings, use_mel_codes_as_input=use_mel_codes_as_input,
                   checkpointing=checkpointing, types=types)
        self.cfg = resolve_config(cfg)
        for k in ("number_text_tokens", "start_text_token", "stop_text_token", "number_mel_codes", "start_mel_token",
                  "stop_mel_token", "layers", "heads", "max_mel_tokens", "max_text_tokens", "model_dim",
                  "max_conditioning_inputs", "mel_length_compression"):
            setattr(self, k, self.cfg[k])
        dev = torch.device(device if device is not None else "cuda")
        self.__dict__["engine"] = GptEngine(self.cfg, dev, dropout_p=dropout_p, seed=seed)
        eng = self.engine
        # parameters = views into the arena, registered under the reference's dotted names
        for key, _ in eng.spec:
            node = self
            parts = key.split(".")
            for part in parts[:-1]:
                if not hasattr(node, part):
                    node.add_module(part, _Node())
                node = getattr(node, part)
            node.register_parameter(parts[-1], nn.Parameter(eng.view(eng.params, key)))
        self.reset_parameters()
        self.attach_grads()
        self.register_load_state_dict_post_hook(lambda mod, incompatible: mod.engine.refresh_shadows())
        self.__dict__["_anchor"] = torch.zeros((), device=dev, requires_grad=True)

    def reset_parameters(self):
        """Reference initialisation: GPT-2 init (normal 0.02, c_proj scaled by 1/sqrt(2L) in HF >= 4), embeddings
        normal(0, .02) (model.py:351-356,234), LayerNorm 1/0, nn.Linear heads default uniform."""
        eng = self.engine
        L = self.cfg["layers"]
        with torch.no_grad():
            for key, shp in eng.spec:
                p = eng.view(eng.params, key)
                if ".ln_" in key or key.startswith("final_norm") or key.startswith("gpt.ln_f"):
                    p.fill_(1.0 if key.endswith("weight") else 0.0)
                elif key.endswith("bias") and not key.endswith("head.bias"):
                    p.zero_()
                elif key.endswith("head.weight"):
                    nn.init.kaiming_uniform_(p, a=5 ** 0.5)
                elif key.endswith("head.bias"):
                    bound = 1.0 / (shp[0] and self.cfg["model_dim"]) ** 0.5
                    p.uniform_(-bound, bound)
                elif key.endswith("c_proj.weight"):
                    p.normal_(0.0, 0.02 / (2 * L) ** 0.5)
                else:
                    p.normal_(0.0, 0.02)
        eng.refresh_shadows()

    def attach_grads(self):
        eng = self.engine
        for key, p in self.named_parameters():
            g = eng.view(eng.grads, key)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def train(self, mode=True):
        super().train(mode)
        self.engine.training = mode
        return self

    def _apply(self, fn, recurse=True):
        """`.cuda()` / `.to(device)` / `accelerator.prepare(model)` from reference-style callers: a no-op when it asks for
        what the model already is (fp32 parameters on its own GPU); anything that would really move or re-type the
        parameters is refused, because they are views into the engine's arenas."""
        probe = self.engine.params[:1]
        out = fn(probe)
        if out.device != probe.device or out.dtype != probe.dtype:
            raise NotImplementedError("UnifiedVoice lives on the GPU it was created on (%s, fp32 master parameters in flat "
                                      "arenas); construct it with device=... instead of moving / casting it" % probe.device)
        return self

    def zero_grad(self, set_to_none=False):
        """Zeroes the flat gradient arena and keeps the `.grad` views attached (the nn.Module default would drop the views
        and leave stale gradients in the arena)."""
        self.engine.zero_grad()
        self.attach_grads()

    def forward(self, text_inputs, text_lengths, mel_codes, wav_lengths, types=None, text_first=True, raw_mels=None,
                return_attentions=False, return_latent=False, clip_inputs=True):
        if raw_mels is not None or return_attentions or not text_first:
            raise NotImplementedError("ttts_amd.UnifiedVoice implements the training path only "
                                      "(text_first=True, raw_mels=None, return_attentions=False)")
        if types is not None:
            text_inputs = text_inputs * (1 + types).unsqueeze(-1)
        eng = self.engine
        eng.set_tokens(*prepare_tokens(self.cfg, text_inputs, text_lengths, mel_codes, wav_lengths, clip_inputs))
        if return_latent:
            eng.training = self.training
            with torch.no_grad():
                eng.forward()
            B, Tt, Tm = eng._bufs_key
            return eng.b["enc"][B * Tt:].view(B, Tm, -1)[:, :-2].float()
        if torch.is_grad_enabled():
            return _GptStep.apply(self, self._anchor)
        eng.training = self.training
        eng.forward()
        lo = eng.b["losses"]
        return lo[0].clone(), lo[1].clone(), eng.mel_logits()


    # ---- inference (ttts/gpt/model.py:357-396, 533-562) ---------------------------------------------------------------
    def post_init_gpt2_config(self, use_deepspeed=False, kv_cache=False, half=False):
        """The reference builds its HF `GPT2InferenceModel` wrapper here.  This build decodes on `GptDecoder` (always with a
        KV cache, bf16 matmuls as in training); the arguments are accepted for source compatibility and change nothing:
        results follow the reference's cache-less path whatever `kv_cache` says (see gpt/decode.py)."""
        if use_deepspeed:
            raise NotImplementedError("deepspeed inference kernels are CUDA-only and not part of this build")
        self.__dict__["decoder"] = GptDecoder(self.engine)
        return self

    def build_aligned_inputs_and_targets(self, input, start_token, stop_token):
        """model.py:397-400."""
        return F.pad(input, (1, 0), value=start_token), F.pad(input, (0, 1), value=stop_token)

    @torch.no_grad()
    def inference_speech(self, text_inputs, mel_codes, input_tokens=None, num_return_sequences=1, max_generate_length=None,
                         typical_sampling=False, typical_mass=.9, **hf_generate_kwargs):
        """Sample mel codes after the prompt `mel_codes` for `text_inputs` (model.py:533-562).  Supported HF generate
        keywords: do_sample, temperature, top_k, top_p, repetition_penalty, length_penalty (beam-search only in HF:
        accepted and ignored), plus `seed` for the counter-hash sampler.  num_beams > 1 and `input_tokens` continuation
        are not built.  Returns int64 (B * num_return_sequences, <= max length) codes, finished rows padded with
        stop_mel_token."""
        if input_tokens is not None:
            raise NotImplementedError("continuing from input_tokens is not part of this build")
        kw = dict(hf_generate_kwargs)
        if kw.pop("num_beams", 1) != 1:
            raise NotImplementedError("beam search is not part of this build")
        kw.pop("length_penalty", None)
        sampler = dict(do_sample=kw.pop("do_sample", False), temperature=kw.pop("temperature", 1.0) or 1.0,
                       top_k=kw.pop("top_k", 0) or 0, top_p=kw.pop("top_p", 1.0) or 1.0,
                       repetition_penalty=kw.pop("repetition_penalty", 1.0) or 1.0, seed=kw.pop("seed", 0))
        if kw:
            raise TypeError("unsupported generate arguments: %s" % sorted(kw))
        if "decoder" not in self.__dict__:
            self.post_init_gpt2_config()
        dev = self.engine.device
        text_inputs = F.pad(text_inputs.to(dev), (0, 1), value=self.stop_text_token)
        text_inp, _ = self.build_aligned_inputs_and_targets(text_inputs, self.start_text_token, self.stop_text_token)
        mel_inp, _ = self.build_aligned_inputs_and_targets(mel_codes.to(dev), self.start_mel_token, self.stop_mel_token)
        n_new = self.max_mel_tokens - 1 if max_generate_length is None else int(max_generate_length)
        n_new = min(n_new, self.max_mel_tokens + 2 - mel_inp.shape[1])        # the learned position table ends there
        return self.decoder.generate(text_inp.long(), mel_inp.long(), n_new, num_return_sequences=num_return_sequences,
                                     typical_mass=typical_mass if typical_sampling else 0.0, **sampler)


class FusedAdamW:
    """Optimizer facade over GptEngine.optimizer_step with the reference trainer's hyper-parameters
    (AdamW lr 1e-4, betas (0.9, 0.96), weight_decay 0.01 -- ttts/gpt/train.py:56; clip 1.0 :115; warm-up :36-40)."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.96), eps=1e-8, weight_decay=0.01, max_norm=1.0, warmup_steps=500):
        self.engine = model.engine
        self.kw = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_norm=max_norm,
                       warmup_steps=warmup_steps)

    def step(self):
        self.engine.optimizer_step(**self.kw)   # also zeroes the gradient arena
        self.engine.step_count += 1

    def zero_grad(self, set_to_none=False):
        self.engine.zero_grad()

    def grad_norm(self):
        return float(self.engine.opt_state[4])

    def last_lr(self):
        return float(self.engine.opt_state[1])
