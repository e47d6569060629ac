"""Autoregressive decoding on the GPT engine: `inference_speech` of the reference (ttts/gpt/model.py:533-562) =
GPT2InferenceModel.forward (:109-184) driven by transformers' sample loop, re-designed around a KV cache and ONE hipGraph.

  prefill : the engine's training-path forward (eval mode) over [start, text, stop | start_mel, prompt codes]; the K / V
            of every layer go to the cache (`kv_cache_fill`, replicated for num_return_sequences), the last position's
            final-norm state gives the first logits.
  step    : embed(token, position) -> per layer: LN -> c_attn GEMM -> attn_decode (append K / V, attend) -> c_proj GEMM (+resid)
            -> LN -> c_fc GEMM (+GELU) -> c_proj GEMM (+resid) -> ln_f -> final_norm -> mel_head GEMM (fp32 logits)
            -> sample_logits -> decode_advance (for <= 16 sequences the LayerNorms are fused into `linear_decode`, 34 launches).
            All positions are device-side counters: the launches are captured once
            and replayed per token; the host only looks at the `unfinished` counter every `poll` tokens.

Semantics follow the reference's default cache-less path (post_init_gpt2_config(kv_cache=False), api_zh.py:52): the token
at mel index i always gets position embedding i.  (The reference's kv_cache=True branch embeds incremental tokens with
`attention_mask.shape[1] - mel_len`, one position too far, model.py:146-149, and so disagrees with its own cache-less
path; the cache here reproduces the cache-less numbers.)
"""
import torch

from .. import ops
from ..lib import EPI_GELU_BF16, EPI_RESID_ADD_F32, EPI_STORE_F32


class GptDecoder:
    def __init__(self, engine):
        self.eng = engine
        self._key = None
        self._graph = None
        self._graph_key = None

    # ---- buffers ------------------------------------------------------------------------------------------------
    def _ensure(self, M, S_max, steps):
        key = (M, S_max, steps)
        if self._key == key:
            return
        e = self.eng
        c, dev = e.c, e.device
        D, L, H = c["model_dim"], c["layers"], c["heads"]
        dh = D // H
        f32, bf = torch.float32, torch.bfloat16
        b = {}
        b["kc"] = [torch.zeros(M, H, S_max, dh, dtype=bf, device=dev) for _ in range(L)]
        b["vc"] = [torch.zeros(M, H, S_max, dh, dtype=bf, device=dev) for _ in range(L)]
        b["x"] = [torch.zeros(M, D, dtype=f32, device=dev) for _ in range(3)]
        b["ln"] = torch.zeros(M, D, dtype=bf, device=dev)
        b["qkv"] = torch.zeros(M, 3 * D, dtype=bf, device=dev)
        b["att"] = torch.zeros(M, D, dtype=bf, device=dev)
        b["fc_pre"] = torch.zeros(M, 4 * D, dtype=bf, device=dev)
        b["fc_act"] = torch.zeros(M, 4 * D, dtype=bf, device=dev)
        b["lnf"] = torch.zeros(M, D, dtype=f32, device=dev)
        b["enc"] = torch.zeros(M, D, dtype=bf, device=dev)
        b["stats"] = [torch.zeros(M, dtype=f32, device=dev) for _ in range(2)]
        b["logits"] = torch.zeros(M, e.ld_m, dtype=f32, device=dev)
        b["ctr"] = torch.zeros(4, dtype=torch.int32, device=dev)
        b["tokens"] = torch.zeros(M, dtype=torch.int64, device=dev)
        b["finished"] = torch.zeros(M, dtype=torch.uint8, device=dev)
        b["out"] = torch.zeros(M, steps, dtype=torch.int64, device=dev)
        b["history"] = torch.zeros(M, S_max + 1, dtype=torch.int64, device=dev)
        self.b = b
        self._key = key
        self._graph = None

    # ---- one decode step (launch sequence; captured by generate) ----------------------------------------------------
    def _forward_token(self, Tt):
        e, b = self.eng, self.b
        c = e.c
        D, L, H = c["model_dim"], c["layers"], c["heads"]
        dh = D // H
        P = lambda k: e.view(e.params, k)  # noqa: E731
        st = b["stats"]
        x = b["x"]
        ops.decode_embed(b["tokens"], P("mel_embedding.weight"), P("mel_pos_embedding.emb.weight"), b["ctr"], -Tt, x[0])
        M = x[0].shape[0]
        skinny = M <= 16        # M <= 16 rows: LayerNorm-fused 16-column MFMA kernel; more rows: the training path's tile GEMM
        cur = 0
        for i in range(L):
            pre = "gpt.h.%d." % i
            x0, x1, x2 = x[cur], x[(cur + 1) % 3], x[(cur + 2) % 3]
            if skinny:
                ops.linear_decode(x0, e.wT[pre + "attn.c_attn.weight"], b["qkv"], P(pre + "attn.c_attn.bias"),
                                  ln1=(P(pre + "ln_1.weight"), P(pre + "ln_1.bias")))
            else:
                ops.layernorm_fwd(x0, P(pre + "ln_1.weight"), P(pre + "ln_1.bias"), b["ln"], st[0], st[1])
                ops.gemm_nt(b["ln"], e.wT[pre + "attn.c_attn.weight"], b["qkv"], P(pre + "attn.c_attn.bias"))
            ops.attn_decode(b["qkv"], b["kc"][i], b["vc"][i], b["ctr"], b["att"], dh ** -0.5)
            if skinny:
                ops.linear_decode(b["att"], e.wT[pre + "attn.c_proj.weight"], x1, P(pre + "attn.c_proj.bias"),
                                  epilogue=EPI_RESID_ADD_F32, resid=x0)
                ops.linear_decode(x1, e.wT[pre + "mlp.c_fc.weight"], b["fc_act"], P(pre + "mlp.c_fc.bias"),
                                  epilogue=EPI_GELU_BF16, ln1=(P(pre + "ln_2.weight"), P(pre + "ln_2.bias")))
                ops.linear_decode(b["fc_act"], e.wT[pre + "mlp.c_proj.weight"], x2, P(pre + "mlp.c_proj.bias"),
                                  epilogue=EPI_RESID_ADD_F32, resid=x1)
            else:
                ops.gemm_nt(b["att"], e.wT[pre + "attn.c_proj.weight"], x1, P(pre + "attn.c_proj.bias"),
                            epilogue=EPI_RESID_ADD_F32, resid_in=x0)
                ops.layernorm_fwd(x1, P(pre + "ln_2.weight"), P(pre + "ln_2.bias"), b["ln"], st[0], st[1])
                ops.gemm_nt(b["ln"], e.wT[pre + "mlp.c_fc.weight"], b["fc_act"], P(pre + "mlp.c_fc.bias"), aux=b["fc_pre"],
                            epilogue=EPI_GELU_BF16)
                ops.gemm_nt(b["fc_act"], e.wT[pre + "mlp.c_proj.weight"], x2, P(pre + "mlp.c_proj.bias"),
                            epilogue=EPI_RESID_ADD_F32, resid_in=x1)
            cur = (cur + 2) % 3
        if skinny:
            ops.linear_decode(x[cur], e.w("mel_head.weight"), b["logits"], P("mel_head.bias"), epilogue=EPI_STORE_F32,
                              ln1=(P("gpt.ln_f.weight"), P("gpt.ln_f.bias")), ln2=(P("final_norm.weight"), P("final_norm.bias")),
                              n=e.nm)
        else:
            ops.layernorm_fwd(x[cur], P("gpt.ln_f.weight"), P("gpt.ln_f.bias"), b["lnf"], st[0], st[1])
            ops.layernorm_fwd(b["lnf"], P("final_norm.weight"), P("final_norm.bias"), b["enc"], st[0], st[1])
            ops.gemm_nt(b["enc"], e.w("mel_head.weight"), b["logits"], P("mel_head.bias"), n=e.nm, epilogue=EPI_STORE_F32)

    def _sample(self, logits, row_div, hist_base, s, probs_out=None):
        b, c = self.b, self.eng.c
        ops.sample_logits(logits, b["ctr"], b["tokens"], b["finished"], self.eng.nm, history=b["history"], hist_base=hist_base,
                          out=b["out"], row_div=row_div, repetition_penalty=s["repetition_penalty"],
                          typical_mass=s["typical_mass"], temperature=s["temperature"], top_k=s["top_k"], top_p=s["top_p"],
                          do_sample=s["do_sample"], eos_token=c["stop_mel_token"], pad_token=c["stop_mel_token"], seed=s["seed"],
                          probs_out=probs_out)
        ops.decode_advance(b["ctr"], b["finished"])

    # ---- public ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, text_inp, mel_inp, max_new_tokens, num_return_sequences=1, do_sample=False, temperature=1.0, top_k=0,
                 top_p=1.0, repetition_penalty=1.0, typical_mass=0.0, seed=0, capture=True, poll=32, return_logits=False,
                 forced_tokens=None):
        """text_inp (B, Tt) = [start, text, stop]; mel_inp (B, P) = [start_mel, prompt codes]; both int64 on the device.
        Returns codes (B * num_return_sequences, n <= max_new_tokens) int64 -- rows that finished early are padded with
        stop_mel_token, trailing all-pad columns are cut like HF's early stop.
        return_logits: also the fp32 logits of every step (M, steps, classes) (eager launches; tests).
        forced_tokens (M, steps): teacher forcing -- feed these instead of the sampled tokens (tests)."""
        e = self.eng
        c = e.c
        B, Tt = text_inp.shape
        Pn = mel_inp.shape[1]
        rep = int(num_return_sequences)
        M = B * rep
        D, L, H = c["model_dim"], c["layers"], c["heads"]
        dh = D // H
        n_pos = c["max_mel_tokens"] + 2
        if Pn + max_new_tokens > n_pos:
            raise ValueError("prompt (%d) + max_new_tokens (%d) exceeds the %d learned mel positions" % (Pn, max_new_tokens, n_pos))
        S0 = Tt + Pn
        S_max = S0 + max_new_tokens
        self._ensure(M, S_max, max_new_tokens)
        b = self.b
        s = dict(do_sample=bool(do_sample), temperature=float(temperature), top_k=int(top_k or 0), top_p=float(top_p),
                 repetition_penalty=float(repetition_penalty), typical_mass=float(typical_mass or 0.0), seed=int(seed))
        # ---- prefill through the training-path forward (eval: no dropout)
        was_training = e.training
        e.training = False
        dummy_t = torch.zeros_like(text_inp)
        dummy_m = torch.zeros_like(mel_inp)
        e.set_tokens(text_inp, dummy_t, mel_inp, dummy_m)
        e.forward()
        e.training = was_training
        for i in range(L):
            ops.kv_cache_fill(e.b["qkv"][i], b["kc"][i], b["vc"][i], B, S0, H, dh, rep)
        # ctr[0] = sequence index of the token a decode step processes; the prefill's own sample + advance moves it to S0
        b["ctr"].copy_(torch.tensor([S0 - 1, 0, M, 0], dtype=torch.int32), non_blocking=True)
        b["finished"].zero_()
        b["out"].fill_(c["stop_mel_token"])
        hist0 = torch.cat([torch.ones(B, Tt, dtype=torch.int64, device=e.device), mel_inp], dim=1)   # HF's input_ids row
        b["history"].zero_()
        b["history"][:, :S0] = hist0.repeat_interleave(rep, dim=0)
        # first logits: final-norm state of the last prompt position, fp32 head GEMM
        enc_m = e.b["enc"][B * Tt:]
        last = enc_m[Pn - 1::Pn]                                              # (B, D) view, row stride Pn * D
        logits0 = torch.zeros(B, e.ld_m, dtype=torch.float32, device=e.device)
        ops.gemm_nt(last, e.w("mel_head.weight"), logits0, e.view(e.params, "mel_head.bias"), n=e.nm, epilogue=EPI_STORE_F32)
        all_logits = [logits0.repeat_interleave(rep, dim=0)[:, :e.nm].clone()] if return_logits else None
        self._sample(logits0, rep, S0, s)
        if forced_tokens is not None:
            b["tokens"].copy_(forced_tokens[:, 0])
        eager = return_logits or forced_tokens is not None or not capture
        steps_done = 1
        if not eager:
            gkey = (self._key, Tt, S0, tuple(sorted(s.items())))
            if self._graph is None or self._graph_key != gkey:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                ctr_save = b["ctr"].clone(); fin_save = b["finished"].clone(); tok_save = b["tokens"].clone()
                out_save = b["out"].clone(); hist_save = b["history"].clone()
                with torch.cuda.graph(g):
                    self._forward_token(Tt)
                    self._sample(b["logits"], 1, S0, s)
                # capture only records, but be explicit: the state the recorded step will start from
                b["ctr"].copy_(ctr_save); b["finished"].copy_(fin_save); b["tokens"].copy_(tok_save)
                b["out"].copy_(out_save); b["history"].copy_(hist_save)
                self._graph, self._graph_key = g, gkey
        while steps_done < max_new_tokens:
            if eager:
                self._forward_token(Tt)
                if return_logits:
                    all_logits.append(b["logits"][:, :e.nm].clone())
                self._sample(b["logits"], 1, S0, s)
                if forced_tokens is not None:
                    b["tokens"].copy_(forced_tokens[:, steps_done])
            else:
                self._graph.replay()
            steps_done += 1
            if steps_done % poll == 0 and forced_tokens is None and int(b["ctr"][2].item()) == 0:
                break
        fin = b["finished"].bool()
        out = b["out"][:, :steps_done]
        if forced_tokens is None and bool(fin.all()):
            # HF stops right after the step in which the last row drew eos: cut the all-pad tail the polling ran past
            stop = c["stop_mel_token"]
            first_eos = (out == stop).int().argmax(dim=1)           # every row has one
            out = out[:, :int(first_eos.max().item()) + 1]
        if return_logits:
            return out.clone(), torch.stack(all_logits, dim=1)
        return out.clone()
