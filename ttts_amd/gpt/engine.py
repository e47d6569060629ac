"""GPT train-step engine: the MI355X-native restatement of the reference's hot loop body

    loss_text, loss_mel, mel_logits = UnifiedVoice(...)(text, text_lengths, mel, wav_lengths)    ttts/gpt/model.py:453-510
    loss.backward(); clip_grad_norm_(1.0); AdamW.step(); LambdaLR.step()                          ttts/gpt/train.py:104-120

as a fixed sequence of hand-written HIP kernels (C ABI, `ttts_amd.ops`) over memory laid out once for the GPU:

* ONE flat fp32 arena each for parameters, gradients and the two Adam moments (state-dict tensors are views into
  it, so the 84 `.item()` grad-norm syncs / 84 optimizer launches of the reference collapse to 3 launches);
* a flat bf16 "shadow" of the parameters written by the AdamW kernel, plus transposed bf16 copies of the GEMM
  weights, so that every forward / dX GEMM reads both operands K-contiguously;
* an fp32 residual stream with one saved copy per LayerNorm input (no gradient checkpointing: 288 GB of HBM make the
  reference's recompute-to-save-memory trade unnecessary; results are identical);
* text and mel rows of the final hidden state in a split layout so each head GEMM sees contiguous rows.

The whole step is free of host reads of device data, hence capturable in one hipGraph (`capture=True`).
"""
import collections
import math

import os

import torch

from .. import ops
from ..lib import EPI_DGELU_BF16, EPI_GELU_BF16, EPI_RESID_ADD_F32, EPI_STORE_BF16, TttsError

DEFAULTS = dict(layers=8, model_dim=512, heads=8, max_text_tokens=120, max_mel_tokens=250,
                max_conditioning_inputs=1, mel_length_compression=1024, number_text_tokens=256,
                start_text_token=None, number_mel_codes=8194, start_mel_token=8192, stop_mel_token=8193,
                train_solo_embeddings=False, use_mel_codes_as_input=True, checkpointing=True, types=1)


def resolve_config(cfg):
    """UnifiedVoice constructor arguments with the reference's defaults (ttts/gpt/model.py:293-297,316-318)."""
    c = dict(DEFAULTS)
    c.update(cfg or {})
    if c["start_text_token"] is None:
        c["start_text_token"] = c["number_text_tokens"] * c["types"]
    c["stop_text_token"] = 0
    if not c["use_mel_codes_as_input"] or c["train_solo_embeddings"]:
        raise NotImplementedError("ttts_amd covers the live training path: use_mel_codes_as_input=True, "
                                  "train_solo_embeddings=False (ttts/gpt/config.json:20,28)")
    if c["model_dim"] % c["heads"] or c["model_dim"] // c["heads"] not in (32, 64, 128):
        raise NotImplementedError("head_dim must be 32, 64 or 128")
    return c


def param_spec(c):
    """(key, shape) in the reference's state-dict order (tests/golden/surface.json pins it)."""
    d, L = c["model_dim"], c["layers"]
    nt = c["number_text_tokens"] * c["types"] + 1
    spec = [("text_embedding.weight", (nt, d)), ("mel_embedding.weight", (c["number_mel_codes"], d))]
    for i in range(L):
        p = "gpt.h.%d." % i
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
             ("text_head.weight", (nt, d)), ("text_head.bias", (nt,)),
             ("mel_head.weight", (c["number_mel_codes"], d)), ("mel_head.bias", (c["number_mel_codes"],))]
    return spec


def _up(x, m):
    return (x + m - 1) // m * m


def arena_layout(spec):
    """({key: element offset}, total elements) of the flat parameter / gradient arena: state-dict order, every tensor on an
    8-element granule (fp32 float4 and bf16 16-byte alignment).  Pure host arithmetic (CPU tests restate nothing)."""
    offsets, off = {}, 0
    for k, shp in spec:
        offsets[k] = off
        off += _up(math.prod(shp), 8)
    return offsets, off


def exchange_ranges(offsets, n_arena, layers, split=None):
    """Element ranges of the flat gradient arena that are final after backward part 0 / part 1 (GptEngine.backward):
    (split, [(lo, hi), ...] ready after the heads + layers L-1 .. split, [(lo, hi), ...] ready at the end).  The arena is in
    state-dict order: embeddings, h.0 .. h.L-1, ln_f, position tables, final_norm, heads -- so each group is contiguous, and
    the four ranges tile [0, n_arena) exactly once (tests/test_host_cpu.py)."""
    split = layers // 2 if split is None else split
    h_split = offsets["gpt.h.%d.ln_1.weight" % split]
    pos0, fin0 = offsets["mel_pos_embedding.emb.weight"], offsets["final_norm.weight"]
    return split, [(h_split, pos0), (fin0, n_arena)], [(0, h_split), (pos0, fin0)]


def left_out_problems(tiles, slots):
    """Which problems of a grouped dW launch go to the split-K kernel instead (GptEngine._dw_plan): tiles[j] = 128 x 128 output
    tiles of problem j, slots = resident workgroups (2 per CU).  A total just above a multiple of `slots` would cost a whole,
    mostly empty extra round; the smallest-sum subset of problems that covers the remainder is taken out.  Returns a set of
    indices (empty: everything stays grouped)."""
    total = sum(tiles)
    rem = total % slots
    if total <= slots or rem == 0 or rem >= (slots * 5) // 8:
        return set()
    best = {0: ()}                      # reachable tile sums < rem + max(tiles) -> one subset each (first found)
    for j, t in enumerate(tiles):
        for ssum, sel in list(best.items()):
            if ssum < rem and (ssum + t) not in best:
                best[ssum + t] = sel + (j,)
    cand = [ssum for ssum in best if ssum >= rem]
    if not cand or min(cand) >= total:
        return set()
    return set(best[min(cand)])


class GptEngine:
    """Owns the arenas and activation buffers of one model replica on one GPU."""

    def __init__(self, cfg, device, dropout_p=0.1, seed=0):
        if not torch.cuda.is_available():
            raise TttsError("ttts_amd needs a ROCm GPU: there is no CPU path (the oracle lives in oracle/, for tests)")
        self.c = resolve_config(cfg)
        self.device = torch.device(device)
        self.dropout_p = float(dropout_p)   # HF GPT2Config default embd/attn/resid_pdrop = 0.1
        self.training = True
        self.seed = int(seed)
        self.step_count = 0                 # optimizer steps taken (host mirror)
        self.dw_split_overlap = os.environ.get("TTTS_DW_SPLIT_OVERLAP", "0") == "1"   # grouped dW of the upper half under the lower half's chain
        self.tail_overlap = os.environ.get("TTTS_TAIL_OVERLAP", "0") == "1"   # small end-of-backward launches beside the grouped dW (see _run_dw)
        self.overlap_dw = os.environ.get("TTTS_OVERLAP_DW", "0") == "1"   # dW GEMMs on a side stream (see backward);
        # off by default: measured +0.7 % only, and concurrent kernels blur per-kernel profiles
        # Deferred, grouped weight gradients (see _dw_plan): every layer keeps its four dY buffers and ALL dW GEMMs of a
        # backward section run as one launch at its end, one workgroup per output tile over the whole token dimension --
        # no split-reduction slabs, no reduce kernels.  TTTS_GROUPED_DW=0 restores one split-K dW GEMM per weight.
        self.grouped_dw = os.environ.get("TTTS_GROUPED_DW", "1") == "1"
        # attention projection + residual add + ln_2 in one launch over whole rows (ttts_gemm_nt_resid_ln_bf16; same bits).  OFF by
        # default: 25.1 vs 32.0 us stand-alone, but inside the step 3.393 vs 3.372 ms on the same box (profiles/r04_ab_fused_ln.txt) --
        # its 145 workgroups pull the fp32 residual through 57 % of the CUs, and in the step that stream comes from HBM.
        self.fused_ln = os.environ.get("TTTS_FUSED_LN", "0") == "1"
        self._dw_plans = {}
        self.seed_ctr = torch.zeros(1, dtype=torch.int32, device=self.device)   # this replica's dropout stream counter (device side:
        # graph-replay safe); handed to every dropout-capable kernel call -- the library holds no state of its own
        self.spec = param_spec(self.c)
        self.shapes = dict(self.spec)
        self.offsets, off = arena_layout(self.spec)
        self.n_arena = off
        dev = self.device
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.opt_state = torch.zeros(8, dtype=torch.float32, device=dev)
        self.gn_ws = ops.gradnorm_workspace(off, dev)
        d = self.c["model_dim"]
        self.nt = self.c["number_text_tokens"] * self.c["types"] + 1
        self.nm = self.c["number_mel_codes"]
        # row pitch of the logits / d-logits buffers = K of the heads' dX GEMMs: a multiple of 64 keeps those on the LDS-DMA kernels
        # (mel head dX 8208 x 512 at K 1032: 23.8 us on the register-staged kernel; at K 1088: 17.3 us -- tools/ubench/nt_phase.cpp)
        self.ld_t, self.ld_m = _up(self.nt, 64), _up(self.nm, 64)
        # transposed bf16 shadows: forward B operands of the Conv1D layers, dX B operands of the heads
        self.wT = {}
        entries = []
        for i in range(self.c["layers"]):
            for nm in ("attn.c_attn", "attn.c_proj", "mlp.c_fc", "mlp.c_proj"):
                k = "gpt.h.%d.%s.weight" % (i, nm)
                src = self.view(self.params, k)
                t = torch.zeros(src.shape[1], src.shape[0], dtype=torch.bfloat16, device=dev)
                self.wT[k] = t
                entries.append((self.view(self.shadow, k), t))
        for k, ld in (("text_head.weight", self.ld_t), ("mel_head.weight", self.ld_m)):
            src = self.view(self.params, k)
            t = torch.zeros(d, ld, dtype=torch.bfloat16, device=dev)   # [in, out padded]; pad columns stay zero
            self.wT[k] = t
            entries.append((self.view(self.shadow, k), t[:, :src.shape[0]]))
        self.cast_plan = ops.TransposePlan(entries, dev)   # bf16 shadow -> its transposed copy (half the read traffic of the fp32 master)
        self._bufs_key = None
        self._graph = None
        self._graph_key = None
        # buffers / descriptor tables / captured graphs of the last few batch shapes (see _ensure_buffers)
        self._shape_cache = collections.OrderedDict()
        self._capture_error = None   # one line on why a hipGraph capture was refused (bench.py reports it), else None

    # ---- parameter plumbing -------------------------------------------------------------------------------
    def view(self, arena, key):
        shp = self.shapes[key]
        o = self.offsets[key]
        return arena[o:o + math.prod(shp)].view(shp)

    def w(self, key):
        """bf16 shadow of a parameter in its stored layout."""
        return self.view(self.shadow, key)

    def refresh_shadows(self):
        """After parameters were written from outside (init / load_state_dict)."""
        self.shadow.copy_(self.params)
        self.cast_plan.run()

    def load_state_dict(self, sd):
        missing = [k for k, _ in self.spec if k not in sd]
        if missing:
            raise KeyError("missing keys: %s" % missing[:4])
        with torch.no_grad():
            for k, shp in self.spec:
                if tuple(sd[k].shape) != tuple(shp):
                    raise ValueError("shape mismatch for %s: %s vs %s" % (k, tuple(sd[k].shape), shp))
                self.view(self.params, k).copy_(sd[k].to(device=self.device, dtype=torch.float32))
        self.refresh_shadows()

    def state_dict(self):
        return {k: self.view(self.params, k).detach().clone() for k, _ in self.spec}

    # ---- buffers ----------------------------------------------------------------------------------------------
    def _ensure_buffers(self, B, Tt, Tm):
        key = (B, Tt, Tm)
        if self._bufs_key == key:
            return
        # Real batches change shape from step to step (the reference clips every batch to its own longest text / clip): the
        # buffers, grouped-launch descriptor tables and captured graphs of the last few shapes are kept (TTTS_SHAPE_CACHE, default
        # 4 shapes: 288 GB of HBM is the budget this is sized for), so a shape that comes back costs neither allocations nor the
        # synchronous descriptor uploads nor a re-capture.  Each cached shape holds a full activation set (~3.4 GB at B 8 x S 1156):
        # the cache multiplies activation memory by up to TTTS_SHAPE_CACHE + 1.  It pays for data sources with a few recurring shapes
        # (fixed-length crops, the benchmark); the reference clips every batch to its own longest text / clip, and rounding those
        # lengths up to buckets is NOT done here: the padded positions are STOP targets that the reference's loss mean counts
        # (model.py:508-509 has no ignore_index), so bucketing would change the loss.  Such a run re-allocates per new shape and
        # runs launch by launch (gpt/train.py passes capture=False).
        cache = self._shape_cache
        if self._bufs_key is not None:
            cache[self._bufs_key] = (self.b, self._Mp, self._dw_plans, self._graph, self._graph_key)
            cache.move_to_end(self._bufs_key)
            while len(cache) > max(1, int(os.environ.get("TTTS_SHAPE_CACHE", "4"))):
                cache.popitem(last=False)
        if key in cache:
            self.b, self._Mp, self._dw_plans, self._graph, self._graph_key = cache.pop(key)
            self._bufs_key = key
            return
        c, dev = self.c, self.device
        D, L, H = c["model_dim"], c["layers"], c["heads"]
        S = Tt + Tm
        M = B * S
        Mp = _up(M, 64)   # row padding (kept zero) of every weight-gradient GEMM operand: whole 64-row DMA tiles
        self._Mp = Mp
        f32, bf = torch.float32, torch.bfloat16
        e = lambda *s, dt=bf: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        z = lambda r, c: torch.zeros(Mp, c, dtype=bf, device=dev)[:r]  # noqa: E731  ([M, c] view of a zero-padded buffer)
        b = {}
        b["xs"] = [e(M, D, dt=f32) for _ in range(2 * L + 1)]
        b["ln1"] = [z(M, D) for _ in range(L)]
        b["ln2"] = [z(M, D) for _ in range(L)]
        b["stats"] = [[e(M, dt=f32) for _ in range(4)] for _ in range(L)]   # mean1, rstd1, mean2, rstd2
        b["qkv"] = [e(M, 3 * D) for _ in range(L)]
        b["att"] = [z(M, D) for _ in range(L)]
        b["lse"] = [e(B * H * S, dt=f32) for _ in range(L)]
        b["fc_pre"] = [e(M, 4 * D) for _ in range(L)]
        b["fc_act"] = [z(M, 4 * D) for _ in range(L)]
        b["lnf"] = e(M, D, dt=f32)
        b["fstats"] = [e(M, dt=f32) for _ in range(4)]
        # (split layout: text rows, then mel rows.  Zero-filled with 64 spare rows, and the dlogits buffers below are zero-padded to
        # whole 64-row tiles: the two head weight gradients ride in the grouped dW launch, whose reduction runs over padded rows)
        b["enc"] = torch.zeros(M + 64, D, dtype=bf, device=dev)[:M]
        b["logits_t"] = torch.zeros(B * Tt, self.ld_t, dtype=bf, device=dev)
        b["logits_m"] = torch.zeros(B * Tm, self.ld_m, dtype=bf, device=dev)
        b["rows_t"] = [e(B * Tt, dt=f32) for _ in range(2)]  # row_loss, row_lse
        b["rows_m"] = [e(B * Tm, dt=f32) for _ in range(2)]
        b["losses"] = torch.zeros(2, dtype=f32, device=dev)  # loss_text, loss_mel
        # backward temporaries
        b["dlog_t"] = torch.zeros(_up(B * Tt, 64), self.ld_t, dtype=bf, device=dev)[:B * Tt]
        b["dlog_m"] = torch.zeros(_up(B * Tm, 64), self.ld_m, dtype=bf, device=dev)[:B * Tm]
        b["d_enc"] = e(M, D)
        b["d_tmp"] = e(M, D, dt=f32)
        b["dres"] = e(M, D, dt=f32)
        b["dres_bf"] = z(M, D)
        b["d_fc"] = z(M, 4 * D)
        b["d_ln"] = e(M, D)
        b["d_att"] = e(M, D)
        b["dqkv"] = z(M, 3 * D)
        b["delta"] = e(B * H * S, dt=f32)
        if self.grouped_dw:   # per-layer dY buffers (0.5 GB at the BASELINE shape): nothing is overwritten before the dW launch
            b["dy_mlp"] = [z(M, D) for _ in range(L)]        # gradient entering mlp.c_proj  (was: dres_bf)
            b["dy_att"] = [z(M, D) for _ in range(L)]        # gradient entering attn.c_proj (was: dres_bf)
            b["d_fc_l"] = [z(M, 4 * D) for _ in range(L)]    # (was: d_fc)
            b["dqkv_l"] = [z(M, 3 * D) for _ in range(L)]    # (was: dqkv)
        b["ln_ws"] = ops.layernorm_bwd_workspace(M, D, dev)
        if self.grouped_dw:   # one workspace per LayerNorm backward: their parameter-gradient sums are finished in ONE launch at the
            b["ln_ws_l"] = [ops.layernorm_bwd_workspace(M, D, dev) for _ in range(2 * L + 2)]   # end of the section (_run_dw)
        tn_shapes = [(D, 3 * D, Mp), (D, D, Mp), (D, 4 * D, Mp), (4 * D, D, Mp), (self.nt, D, B * Tt), (self.nm, D, B * Tm)]
        b["tn_ws"] = max((ops.gemm_tn_workspace(mo, no, kr, dev) for mo, no, kr in tn_shapes), key=lambda t: t.numel())
        # static token buffers (graph replay reads them)
        i64 = torch.int64
        b["text_inp"] = torch.zeros(B, Tt, dtype=i64, device=dev)
        b["text_tar"] = torch.zeros(B * Tt, dtype=i64, device=dev)
        b["mel_inp"] = torch.zeros(B, Tm, dtype=i64, device=dev)
        b["mel_tar"] = torch.zeros(B * Tm, dtype=i64, device=dev)
        self.b = b
        self._bufs_key = key
        self._graph = None
        self._dw_plans = {}
        if self.grouped_dw:   # descriptor tables are built here, outside any graph capture (they upload a small table)
            split = L // 2
            for lo, hi, heads in {(0, L, True), (split, L, True), (0, split, False)}:
                if hi > lo:
                    self._dw_plan(lo, hi, heads)

    def _nt(self, a, w, c, *args, **kw):
        return ops.gemm_nt(a, w, c, *args, **kw)

    def _dw_plan(self, lo, hi, heads):
        """Grouped dW launch for layers lo .. hi-1 (+ the two heads and the final norms when `heads`: the backward section that
        ran _backward_head): (TnPlan | None, [problems left to the split-K path]).
        A problem = (at, bt, grad view).  The grouped kernel gives every 128 x 128 output tile of every problem to one
        workgroup; with 2 workgroups per CU resident, a tile count just above a multiple of 2 x CUs would cost a whole
        extra, mostly empty round (1152 tiles on 512 slots: 2.25 -> 3 rounds).  So the smallest set of problems that
        covers the remainder is taken out and run through the split-K kernel instead."""
        key = (lo, hi, bool(heads))
        if key in self._dw_plans:
            return self._dw_plans[key]
        if torch.cuda.is_current_stream_capturing():
            raise TttsError("grouped dW plan for layers %d..%d must be built before graph capture (only split = layers // 2 "
                            "is prepared by _ensure_buffers)" % (lo, hi))
        b = self.b
        G = lambda k: self.view(self.grads, k)   # noqa: E731
        P_ = self._padded
        probs = []
        for i in reversed(range(lo, hi)):
            pre = "gpt.h.%d." % i
            probs += [(P_(b["ln2"][i]), P_(b["d_fc_l"][i]), G(pre + "mlp.c_fc.weight")),
                      (P_(b["fc_act"][i]), P_(b["dy_mlp"][i]), G(pre + "mlp.c_proj.weight")),
                      (P_(b["ln1"][i]), P_(b["dqkv_l"][i]), G(pre + "attn.c_attn.weight")),
                      (P_(b["att"][i]), P_(b["dy_att"][i]), G(pre + "attn.c_proj.weight"))]
        if heads:                         # the two head weight gradients (reduction over the text / mel rows of the split layout)
            rows64 = lambda t, r: torch.as_strided(t, (_up(r, 64), t.shape[1]), t.stride(), t.storage_offset())   # noqa: E731
            Bt_, Tt_, Tm_ = self._bufs_key
            nt_rows, nm_rows = Bt_ * Tt_, Bt_ * Tm_
            probs += [(rows64(b["dlog_t"], nt_rows)[:, :self.nt], rows64(b["enc"][:nt_rows], nt_rows), G("text_head.weight")),
                      (rows64(b["dlog_m"], nm_rows)[:, :self.nm], rows64(b["enc"][nt_rows:], nm_rows), G("mel_head.weight"))]
        tiles = [ops.tn_desc_tiles(at.shape[1], bt.shape[1]) for at, bt, _ in probs]
        if not hasattr(self, "_cus"):
            self._cus = ops.device_info()["cus"]
        out = left_out_problems(tiles, 2 * self._cus)
        grouped = [pr for j, pr in enumerate(probs) if j not in out]
        single = [pr for j, pr in enumerate(probs) if j in out]
        # one launch takes at most 64 descriptors (the kernel finds its problem with one lane per descriptor): models deeper than
        # 16 layers (the reference constructor accepts e.g. 30) run their grouped problems in chunks
        plans = [ops.TnPlan(grouped[j:j + ops.TN_GROUP_MAX], self.device) for j in range(0, len(grouped), ops.TN_GROUP_MAX)]
        # the other reductions over the rows that only feed the optimizer, also one launch each per section: LayerNorm dgamma /
        # dbeta (+ the bias gradient of the projection consuming the bf16 copy) from the per-call partial sums, and the bias
        # gradients that are plain column sums of a kept dY buffer
        L = self.c["layers"]
        ln, cs = [], []
        if heads:
            ln += [(b["ln_ws_l"][2 * L], G("final_norm.weight"), G("final_norm.bias"), None),
                   (b["ln_ws_l"][2 * L + 1], G("gpt.ln_f.weight"), G("gpt.ln_f.bias"), G("gpt.h.%d.mlp.c_proj.bias" % (L - 1)))]
            cs += [(b["dlog_t"], G("text_head.bias"), self.nt), (b["dlog_m"], G("mel_head.bias"), self.nm)]
        for i in reversed(range(lo, hi)):
            pre = "gpt.h.%d." % i
            ln += [(b["ln_ws_l"][2 * i + 1], G(pre + "ln_2.weight"), G(pre + "ln_2.bias"), G(pre + "attn.c_proj.bias")),
                   (b["ln_ws_l"][2 * i], G(pre + "ln_1.weight"), G(pre + "ln_1.bias"),
                    G("gpt.h.%d.mlp.c_proj.bias" % (i - 1)) if i > 0 else None)]
            # (mlp.c_fc.bias rides in the dGELU epilogue.  The same on the attention backward's dq / dk / dv stores was measured
            # and removed: 320 fp32 atomics per bias element from 2560 waves cost the two kernels +50 us each; so was computing delta in
            # the dQ kernel's prologue instead of the 7 us pre-pass: the dQ kernel, then first to touch dO / O, lost 9 us; and running dQ on a
            # side stream beside dK / dV (they only share the delta pre-pass): GPT step 3.60 vs 3.47 ms -- the two kernels' workgroups
            # crowd each other out of the CUs instead of filling each other's tails)
            cs += [(b["dqkv_l"][i], G(pre + "attn.c_attn.bias"), None)]
        M, D = b["d_fc_l"][0].shape[0], self.c["model_dim"]
        ln_plan = ops.LnFinalizePlan(ln, M, D, self.device)
        cs_plans = [ops.ColsumPlan(cs[j:j + 64], self.device) for j in range(0, len(cs), 64)]   # (may be empty)
        self._dw_plans[key] = (plans, single, ln_plan, cs_plans)
        return self._dw_plans[key]

    def _run_dw(self, lo, hi, heads, tail=None):
        """The section's weight gradients (grouped launches), LayerNorm parameter gradients and bias column sums.  `tail`: a callable
        with more small independent work of the section (the embedding backward).  With tail_overlap the small launches -- each a
        few dozen workgroups, 30 + 39 + 24 us run one after the other -- go to the side stream and run beside the grouped
        weight-gradient launch and beside each other (all four write disjoint gradient rows and only read the section's buffers)."""
        plans, single, ln_plan, cs_plans = self._dw_plan(lo, hi, heads)

        def small():
            ln_plan.run()
            for plan in cs_plans:
                plan.run()
            if tail is not None:
                tail()
        if self.tail_overlap:
            main, side = torch.cuda.current_stream(), self._side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                small()
        for plan in plans:
            plan.run()
        for at, bt, g in single:
            ops.gemm_tn_accum(at, bt, g, workspace=self.b["tn_ws"])
        if self.tail_overlap:
            main.wait_stream(side)
        else:
            small()

    def _padded(self, t):
        """The zero-row-padded [Mp, c] buffer behind an [M, c] activation view (weight-gradient GEMM operand)."""
        return torch.as_strided(t, (self._Mp, t.shape[1]), t.stride(), t.storage_offset())

    def set_tokens(self, text_inp, text_tar, mel_inp, mel_tar):
        B, Tt = text_inp.shape
        Tm = mel_inp.shape[1]
        if Tt > self.c["max_text_tokens"] + 2 or Tm > self.c["max_mel_tokens"] + 2:
            raise ValueError("sequence exceeds the learned position tables (text %d, mel %d)" % (Tt, Tm))
        self._ensure_buffers(B, Tt, Tm)
        b = self.b
        b["text_inp"].copy_(text_inp, non_blocking=True)
        b["text_tar"].copy_(text_tar.reshape(-1), non_blocking=True)
        b["mel_inp"].copy_(mel_inp, non_blocking=True)
        b["mel_tar"].copy_(mel_tar.reshape(-1), non_blocking=True)

    def set_tokens_raw(self, text_inputs, text_lengths, mel_codes, wav_lengths, clip_inputs=True):
        """model.prepare_tokens + set_tokens in ONE kernel launch (ttts_gpt_prepare_tokens): clip to the batch maximum,
        mel padding -> STOP, START / STOP framing, straight into the engine's static token buffers
        (ttts/gpt/model.py:474-489,397-414).  text_inputs / mel_codes: int64 on this GPU; the lengths are host tensors (a
        GPU tensor costs the same two host reads the reference makes, model.py:477,479)."""
        c = self.c
        comp = c["mel_length_compression"]
        tl = text_lengths.tolist() if torch.is_tensor(text_lengths) else list(text_lengths)
        wl = wav_lengths.tolist() if torch.is_tensor(wav_lengths) else list(wav_lengths)
        B = text_inputs.shape[0]
        Tt, Tm = text_inputs.shape[1], mel_codes.shape[1]
        if clip_inputs:
            Tt, Tm = min(Tt, int(max(tl))), min(Tm, int(max(wl)) // comp)
        if Tt + 2 > c["max_text_tokens"] + 2 or Tm + 2 > c["max_mel_tokens"] + 2:
            raise ValueError("sequence exceeds the learned position tables (text %d, mel %d)" % (Tt + 2, Tm + 2))
        self._ensure_buffers(B, Tt + 2, Tm + 2)
        b = self.b
        ops.gpt_prepare_tokens(text_inputs, mel_codes, [int(w) // comp + 1 for w in wl], Tt, Tm, c["start_text_token"],
                               c["stop_text_token"], c["start_mel_token"], c["stop_mel_token"], b["text_inp"], b["text_tar"],
                               b["mel_inp"], b["mel_tar"])

    # ---- dropout seeds: one stream per (step, site) ------------------------------------------------------------
    def _p(self):
        return self.dropout_p if self.training else 0.0

    def _seed(self, site):
        # per-site constant; the per-step variation comes from the device counter the kernels add at run time
        return (self.seed * 0x9E3779B97F4A7C15 + site * 0x632BE59BD9B4E019 + 0x1234567) & (2 ** 64 - 1)

    # ---- forward ---------------------------------------------------------------------------------------------
    def forward(self):
        """Runs on the tokens last given to set_tokens(); fills b['losses'], b['logits_*']."""
        c, b = self.c, self.b
        B, Tt, Tm = self._bufs_key
        D, L, H = c["model_dim"], c["layers"], c["heads"]
        S, dh = Tt + Tm, D // H
        p = self._p()
        P = lambda k: self.view(self.params, k)  # noqa: E731
        ops.embed_fwd(b["text_inp"], b["mel_inp"], P("text_embedding.weight"), P("text_pos_embedding.emb.weight"),
                      P("mel_embedding.weight"), P("mel_pos_embedding.emb.weight"), b["xs"][0], p, self._seed(1), counter=self.seed_ctr)
        for i in range(L):
            pre = "gpt.h.%d." % i
            x0, x1, x2 = b["xs"][2 * i], b["xs"][2 * i + 1], b["xs"][2 * i + 2]
            st = b["stats"][i]
            ops.layernorm_fwd(x0, P(pre + "ln_1.weight"), P(pre + "ln_1.bias"), b["ln1"][i], st[0], st[1])
            self._nt(b["ln1"][i], self.wT[pre + "attn.c_attn.weight"], b["qkv"][i], P(pre + "attn.c_attn.bias"))
            qkv = b["qkv"][i]
            ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], b["att"][i], b["lse"][i], B, H, S, dh, (S * 3 * D, 3 * D),
                         (S * D, D), dh ** -0.5, p, self._seed(16 * i + 2), counter=self.seed_ctr)
            if D == 512 and self.fused_ln:
                # opt-in: the attention projection, the residual add and ln_2 as ONE launch over whole 512-column rows (bit-identical
                # to the two launches below; csrc/gemm.hip gemm_nt_rowln_kernel).  The MLP projection (K = 2048) is never fused:
                # 51.5 vs 50.3 us even stand-alone -- its weight stream is 4 x longer and runs on 145 of the 256 CUs
                ops.gemm_nt_resid_ln(b["att"][i], self.wT[pre + "attn.c_proj.weight"], x1, P(pre + "ln_2.weight"), P(pre + "ln_2.bias"),
                                     b["ln2"][i], st[2], st[3], bias=P(pre + "attn.c_proj.bias"), resid_in=x0, dropout_p=p,
                                     seed=self._seed(16 * i + 3), counter=self.seed_ctr)
            else:
                self._nt(b["att"][i], self.wT[pre + "attn.c_proj.weight"], x1, P(pre + "attn.c_proj.bias"),
                         epilogue=EPI_RESID_ADD_F32, resid_in=x0, dropout_p=p, seed=self._seed(16 * i + 3), counter=self.seed_ctr)
                ops.layernorm_fwd(x1, P(pre + "ln_2.weight"), P(pre + "ln_2.bias"), b["ln2"][i], st[2], st[3])
            self._nt(b["ln2"][i], self.wT[pre + "mlp.c_fc.weight"], b["fc_act"][i], P(pre + "mlp.c_fc.bias"),
                     aux=b["fc_pre"][i], epilogue=EPI_GELU_BF16)
            self._nt(b["fc_act"][i], self.wT[pre + "mlp.c_proj.weight"], x2, P(pre + "mlp.c_proj.bias"),
                     epilogue=EPI_RESID_ADD_F32, resid_in=x1, dropout_p=p, seed=self._seed(16 * i + 4), counter=self.seed_ctr)
        fs = b["fstats"]
        ops.layernorm_fwd(b["xs"][2 * L], P("gpt.ln_f.weight"), P("gpt.ln_f.bias"), b["lnf"], fs[0], fs[1])
        ops.layernorm_fwd(b["lnf"], P("final_norm.weight"), P("final_norm.bias"), b["enc"], fs[2], fs[3],
                          split=(S, Tt))
        enc_t, enc_m = b["enc"][:B * Tt], b["enc"][B * Tt:]
        ops.gemm_nt(enc_t, self.w("text_head.weight"), b["logits_t"], P("text_head.bias"), n=self.nt)
        ops.gemm_nt(enc_m, self.w("mel_head.weight"), b["logits_m"], P("mel_head.bias"), n=self.nm)
        ops.ce_fwd(b["logits_t"], b["text_tar"], b["rows_t"][0], b["rows_t"][1], b["losses"][0:1], self.nt)
        ops.ce_fwd(b["logits_m"], b["mel_tar"], b["rows_m"][0], b["rows_m"][1], b["losses"][1:2], self.nm)

    # ---- backward ----------------------------------------------------------------------------------------------
    def backward(self, w_text=0.01, w_mel=1.0, g_text_dev=None, g_mel_dev=None, part=None, split=None):
        """Accumulates d(w_text*loss_text + w_mel*loss_mel) into self.grads (optionally scaled by device scalars).
        part / split: run only a section of the chain so that a data-parallel exchange of the finished gradients can overlap
        the rest -- part 0 = heads, final norms and layers L-1 .. split; part 1 = layers split-1 .. 0 and the embeddings
        (all state between the two lives in the activation buffers)."""
        c, b = self.c, self.b
        B, Tt, Tm = self._bufs_key
        D, L, H = c["model_dim"], c["layers"], c["heads"]
        S, dh = Tt + Tm, D // H
        p = self._p()
        P = lambda k: self.view(self.params, k)  # noqa: E731
        G = lambda k: self.view(self.grads, k)   # noqa: E731
        # Weight-gradient GEMMs (dW = X^T dY, + bias column sums) run on a SIDE stream: they only feed the optimizer, so they
        # overlap with the data-gradient chain (dX GEMMs, attention backward, LayerNorm backward) on the main stream and
        # fill the CUs that chain leaves idle (292-tile GEMMs on 256 CUs, causal tails, store phases).  Events order the
        # reuse of the scratch buffers (dres_bf, d_fc, dqkv) between the two streams; both streams are captured in the graph.
        main = torch.cuda.current_stream()
        side = self._side_stream() if (self.overlap_dw and part is None and not self.grouped_dw) else main
        if part is not None and not 0 < split < L:
            raise ValueError("backward(part=%r): split must lie strictly inside the layer range (got %r of %d layers)" % (part, split, L))
        lo_layer = 0 if part in (None, 1) else split
        hi_layer = L if part in (None, 0) else split

        def fork():                      # side waits for everything issued on main so far
            if side is not main:
                side.wait_stream(main)

        def done():                      # marker on the side stream that main can wait on before overwriting an operand
            if side is main:
                return None
            ev = torch.cuda.Event()
            ev.record(side)
            return ev

        def wait(ev):
            if ev is not None:
                main.wait_event(ev)

        if part is None and self.grouped_dw and self.dw_split_overlap and L >= 2 and (L // 2, L, True) in self._dw_plans:
            # Grouped weight gradients of the UPPER half of the layers (+ heads) on a side stream while the main stream runs the
            # data-gradient chain of the lower half: the section's batched launches read per-layer buffers only, and the chain's
            # 292-tile GEMMs / causal tails / row kernels leave CUs for them.  The lower half's weight gradients follow on the
            # main stream after the join (they share the split-K workspace and the gradient arena's LayerNorm rows).
            half = L // 2
            self._backward_head(w_text, w_mel, g_text_dev, g_mel_dev, main, fork)
            for i in reversed(range(half, L)):
                self._backward_layer(i, main, fork, done, wait, None, None)
            dws = self._side_stream()
            dws.wait_stream(main)
            with torch.cuda.stream(dws):
                self._run_dw(half, L, True)
            for i in reversed(range(0, half)):
                self._backward_layer(i, main, fork, done, wait, None, None)
            ops.embed_bwd(b["text_inp"], b["mel_inp"], b["dres"], G("text_embedding.weight"),
                          G("text_pos_embedding.emb.weight"), G("mel_embedding.weight"), G("mel_pos_embedding.emb.weight"),
                          p, self._seed(1), counter=self.seed_ctr)
            main.wait_stream(dws)
            self._run_dw(0, half, False)
            return
        if part in (None, 0):
            self._backward_head(w_text, w_mel, g_text_dev, g_mel_dev, side, fork)
        ev_fc = ev_qkv = None            # side-stream reads of d_fc / dqkv by the previous layer
        for i in reversed(range(lo_layer, hi_layer)):
            ev_fc, ev_qkv = self._backward_layer(i, side, fork, done, wait, ev_fc, ev_qkv)
        def embed_tail():
            ops.embed_bwd(b["text_inp"], b["mel_inp"], b["dres"], G("text_embedding.weight"),
                          G("text_pos_embedding.emb.weight"), G("mel_embedding.weight"), G("mel_pos_embedding.emb.weight"),
                          p, self._seed(1), counter=self.seed_ctr)
        ran_dw = self.grouped_dw and (hi_layer > lo_layer or part in (None, 0))
        if ran_dw:      # (the head / final-norm gradients belong to the section that ran _backward_head)
            self._run_dw(lo_layer, hi_layer, part in (None, 0), tail=embed_tail if part in (None, 1) else None)
        if part in (None, 1) and not ran_dw:
            embed_tail()
        if side is not main:
            main.wait_stream(side)       # join: the optimizer / all-reduce needs every dW

    def _backward_head(self, w_text, w_mel, g_text_dev, g_mel_dev, side, fork):
        c, b = self.c, self.b
        B, Tt, Tm = self._bufs_key
        L = c["layers"]
        S = Tt + Tm
        p = self._p()
        P = lambda k: self.view(self.params, k)  # noqa: E731
        G = lambda k: self.view(self.grads, k)   # noqa: E731
        ops.ce_bwd(b["logits_t"], b["text_tar"], b["rows_t"][1], b["dlog_t"], self.nt, w_text, g_text_dev)
        ops.ce_bwd(b["logits_m"], b["mel_tar"], b["rows_m"][1], b["dlog_m"], self.nm, w_mel, g_mel_dev)
        enc_t, enc_m = b["enc"][:B * Tt], b["enc"][B * Tt:]
        fork()
        with torch.cuda.stream(side):
            if not self.grouped_dw:      # grouped: the head weight gradients are two more problems of the section's grouped launch
                ops.gemm_tn_accum(b["dlog_t"], enc_t, G("text_head.weight"), mo=self.nt, workspace=b["tn_ws"])
                ops.gemm_tn_accum(b["dlog_m"], enc_m, G("mel_head.weight"), mo=self.nm, workspace=b["tn_ws"])
            if not self.grouped_dw:      # grouped: part of the section's batched column-sum launch (_run_dw)
                ops.colsum_accum(b["dlog_t"], G("text_head.bias"), n=self.nt)
                ops.colsum_accum(b["dlog_m"], G("mel_head.bias"), n=self.nm)
        ops.gemm_nt(b["dlog_t"], self.wT["text_head.weight"], b["d_enc"][:B * Tt])
        ops.gemm_nt(b["dlog_m"], self.wT["mel_head.weight"], b["d_enc"][B * Tt:])
        fs = b["fstats"]
        gd = self.grouped_dw             # grouped: dgamma / dbeta / dcolsum are finished by the section's batched launch (_run_dw)
        ops.layernorm_bwd(b["d_enc"], b["lnf"], P("final_norm.weight"), fs[2], fs[3], None, b["d_tmp"], None,
                          None if gd else G("final_norm.weight"), None if gd else G("final_norm.bias"),
                          b["ln_ws_l"][2 * L] if gd else b["ln_ws"], split=(S, Tt))
        ops.layernorm_bwd(b["d_tmp"], b["xs"][2 * L], P("gpt.ln_f.weight"), fs[0], fs[1], None, b["dres"],
                          b["dy_mlp"][L - 1] if gd else b["dres_bf"],
                          None if gd else G("gpt.ln_f.weight"), None if gd else G("gpt.ln_f.bias"),
                          b["ln_ws_l"][2 * L + 1] if gd else b["ln_ws"], dropout_p=p, seed=self._seed(16 * (L - 1) + 4),
                          dcolsum=None if gd else G("gpt.h.%d.mlp.c_proj.bias" % (L - 1)), counter=self.seed_ctr)

    def _backward_layer(self, i, side, fork, done, wait, ev_fc, ev_qkv):
        c, b = self.c, self.b
        B, Tt, Tm = self._bufs_key
        D, H = c["model_dim"], c["heads"]
        S, dh = Tt + Tm, D // H
        p = self._p()
        P = lambda k: self.view(self.params, k)  # noqa: E731
        G = lambda k: self.view(self.grads, k)   # noqa: E731
        pre = "gpt.h.%d." % i
        st = b["stats"][i]
        x0, x1 = b["xs"][2 * i], b["xs"][2 * i + 1]
        if self.grouped_dw:
            return self._backward_layer_grouped(i)
        dy = b["dres_bf"]                                  # gradient entering mlp.c_proj (resid dropout applied)
        fork()
        with torch.cuda.stream(side):
            ops.gemm_tn_accum(self._padded(b["fc_act"][i]), self._padded(dy), G(pre + "mlp.c_proj.weight"), workspace=b["tn_ws"])
        ev_dy = done()
        wait(ev_fc)                                        # the previous layer's dW c_fc has consumed d_fc
        self._nt(dy, self.w(pre + "mlp.c_proj.weight"), b["d_fc"], aux=b["fc_pre"][i], epilogue=EPI_DGELU_BF16)
        fork()
        with torch.cuda.stream(side):
            ops.gemm_tn_accum(self._padded(b["ln2"][i]), self._padded(b["d_fc"]), G(pre + "mlp.c_fc.weight"), workspace=b["tn_ws"])
            ops.colsum_accum(b["d_fc"], G(pre + "mlp.c_fc.bias"))
        ev_fc = done()
        self._nt(b["d_fc"], self.w(pre + "mlp.c_fc.weight"), b["d_ln"])
        wait(ev_dy)                                        # dres_bf is rewritten by the LayerNorm backward below
        ops.layernorm_bwd(b["d_ln"], x1, P(pre + "ln_2.weight"), st[2], st[3], b["dres"], b["dres"], b["dres_bf"],
                          G(pre + "ln_2.weight"), G(pre + "ln_2.bias"), b["ln_ws"], dropout_p=p,
                          seed=self._seed(16 * i + 3), dcolsum=G(pre + "attn.c_proj.bias"), counter=self.seed_ctr)
        dy = b["dres_bf"]                                  # gradient entering attn.c_proj
        fork()
        with torch.cuda.stream(side):
            ops.gemm_tn_accum(self._padded(b["att"][i]), self._padded(dy), G(pre + "attn.c_proj.weight"), workspace=b["tn_ws"])
        ev_dy = done()
        self._nt(dy, self.w(pre + "attn.c_proj.weight"), b["d_att"])
        qkv, dqkv = b["qkv"][i], b["dqkv"]
        wait(ev_qkv)                                       # the previous layer's dW c_attn has consumed dqkv
        ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], b["att"][i], b["d_att"], b["lse"][i], dqkv, dqkv[:, D:],
                     dqkv[:, 2 * D:], b["delta"], B, H, S, dh, (S * 3 * D, 3 * D), (S * D, D), dh ** -0.5, p,
                     self._seed(16 * i + 2), counter=self.seed_ctr)
        fork()
        with torch.cuda.stream(side):
            ops.gemm_tn_accum(self._padded(b["ln1"][i]), self._padded(dqkv), G(pre + "attn.c_attn.weight"), workspace=b["tn_ws"])
            ops.colsum_accum(dqkv, G(pre + "attn.c_attn.bias"))
        ev_qkv = done()
        self._nt(dqkv, self.w(pre + "attn.c_attn.weight"), b["d_ln"])
        wait(ev_dy)
        ops.layernorm_bwd(b["d_ln"], x0, P(pre + "ln_1.weight"), st[0], st[1], b["dres"], b["dres"],
                          b["dres_bf"] if i > 0 else None, G(pre + "ln_1.weight"), G(pre + "ln_1.bias"),
                          b["ln_ws"], dropout_p=p if i > 0 else 0.0, seed=self._seed(16 * (i - 1) + 4),
                          dcolsum=G("gpt.h.%d.mlp.c_proj.bias" % (i - 1)) if i > 0 else None, counter=self.seed_ctr)
        return ev_fc, ev_qkv

    def _backward_layer_grouped(self, i):
        """The data-gradient chain of layer i alone: every dY goes to this layer's own buffer and the four weight
        gradients, the bias gradients and the LayerNorm parameter gradients are left to the batched launches at the end of the
        section (_run_dw)."""
        c, b = self.c, self.b
        B, Tt, Tm = self._bufs_key
        D, H = c["model_dim"], c["heads"]
        S, dh = Tt + Tm, D // H
        p = self._p()
        P = lambda k: self.view(self.params, k)  # noqa: E731
        G = lambda k: self.view(self.grads, k)   # noqa: E731
        pre = "gpt.h.%d." % i
        st = b["stats"][i]
        x0, x1 = b["xs"][2 * i], b["xs"][2 * i + 1]
        dy, d_fc, dy_att, dqkv = b["dy_mlp"][i], b["d_fc_l"][i], b["dy_att"][i], b["dqkv_l"][i]
        self._nt(dy, self.w(pre + "mlp.c_proj.weight"), d_fc, aux=b["fc_pre"][i], epilogue=EPI_DGELU_BF16,
                 colsum=G(pre + "mlp.c_fc.bias"))      # the c_fc bias gradient rides in the dGELU epilogue
        self._nt(d_fc, self.w(pre + "mlp.c_fc.weight"), b["d_ln"])
        ops.layernorm_bwd(b["d_ln"], x1, P(pre + "ln_2.weight"), st[2], st[3], b["dres"], b["dres"], dy_att,
                          None, None, b["ln_ws_l"][2 * i + 1], dropout_p=p, seed=self._seed(16 * i + 3), counter=self.seed_ctr)
        self._nt(dy_att, self.w(pre + "attn.c_proj.weight"), b["d_att"])
        qkv = b["qkv"][i]
        ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], b["att"][i], b["d_att"], b["lse"][i], dqkv, dqkv[:, D:],
                     dqkv[:, 2 * D:], b["delta"], B, H, S, dh, (S * 3 * D, 3 * D), (S * D, D), dh ** -0.5, p,
                     self._seed(16 * i + 2), counter=self.seed_ctr)
        self._nt(dqkv, self.w(pre + "attn.c_attn.weight"), b["d_ln"])
        ops.layernorm_bwd(b["d_ln"], x0, P(pre + "ln_1.weight"), st[0], st[1], b["dres"], b["dres"],
                          b["dy_mlp"][i - 1] if i > 0 else None, None, None, b["ln_ws_l"][2 * i],
                          dropout_p=p if i > 0 else 0.0, seed=self._seed(16 * (i - 1) + 4), counter=self.seed_ctr)
        return None, None

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    # ---- optimizer ---------------------------------------------------------------------------------------------
    def optimizer_step(self, lr=1e-4, betas=(0.9, 0.96), eps=1e-8, weight_decay=0.01, max_norm=1.0, warmup_steps=500):
        """get_grad_norm + clip_grad_norm_(1.0) + AdamW.step + zero_grad + LambdaLR(warmup).step
        (ttts/gpt/train.py:114-120) in four launches; refreshes the bf16 shadows."""
        ops.adamw_schedule(self.opt_state, lr, betas[0], betas[1], warmup_steps)
        ops.gradnorm(self.grads, max_norm, self.opt_state, self.gn_ws)
        ops.adamw(self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.shadow, self.opt_state, betas[0],
                  betas[1], eps, weight_decay, zero_grad=True)
        self.cast_plan.run()
        self.seed_ctr.add_(1)   # next step draws fresh dropout masks (also under graph replay)

    def zero_grad(self):
        self.grads.zero_()

    # ---- whole step (optionally replayed from one hipGraph) ---------------------------------------------------
    def grad_exchange_plan(self, split=None):
        """Element ranges of the flat gradient arena that are final after backward part 0 / part 1 (see `backward`):
        ([(lo, hi), ...] ready after the heads + layers L-1 .. split, [(lo, hi), ...] ready at the end).  The arena is in
        state-dict order: embeddings, h.0 .. h.L-1, ln_f, position tables, final_norm, heads."""
        return exchange_ranges(self.offsets, self.n_arena, self.c["layers"], split)

    def train_step(self, tokens, w_text=0.01, w_mel=1.0, capture=False, exchange=None, exchange_range=None, **opt):
        """tokens = (text_inp, text_tar, mel_inp, mel_tar) int64 tensors (see model.prepare_tokens).
        exchange: optional callable run between backward and the optimizer (the data-parallel gradient all-reduce).
        exchange_range: optional callable (lo, hi) -> handle with .wait(): asynchronous all-reduce of grads[lo:hi].  With it
        the backward runs in two sections and the ranges that are final after the first one (upper layers, heads: 46 % of
        the bytes) are exchanged while the second section computes (`grad_exchange_plan`).
        capture=True replays the step from hipGraphs: one graph without an exchange, two (forward+backward | optimizer)
        around `exchange`, three (forward + backward part 0 | backward part 1 | optimizer) around `exchange_range`.
        tokens=None: the token buffers were already filled (set_tokens_raw).
        Returns nothing: losses stay on the device (no host sync in the hot loop)."""
        if tokens is not None:
            self.set_tokens(*tokens)
        mode = "range" if exchange_range is not None else ("whole" if exchange is not None else "none")
        graphs = None
        if capture:
            key = (self._bufs_key, w_text, w_mel, tuple(sorted(opt.items())), mode)
            if self._graph is None or self._graph_key != key:
                self._graph, self._graph_key = self._capture_step(w_text, w_mel, mode, opt), key
            graphs = self._graph if self._graph[0] is not None else None   # None: capture was refused, run launch by launch
        if mode == "range":
            split, first, second = self.grad_exchange_plan()
            if graphs:
                graphs[0].replay()
            else:
                self.forward()
                self.backward(w_text, w_mel, part=0, split=split)
            pending = [exchange_range(lo, hi) for lo, hi in first]
            if graphs:
                graphs[1].replay()
            else:
                self.backward(w_text, w_mel, part=1, split=split)
            pending += [exchange_range(lo, hi) for lo, hi in second]
            for h in pending:
                if h is not None:
                    h.wait()
            if graphs:
                graphs[2].replay()
            else:
                self.optimizer_step(**opt)
        else:
            if graphs:
                graphs[0].replay()   # capture only records; every step (the first included) is a replay
            else:
                self.forward()
                self.backward(w_text, w_mel)
            if exchange is not None:
                exchange()
            if graphs:
                if graphs[1] is not None:
                    graphs[1].replay()
            else:
                self.optimizer_step(**opt)
        self.step_count += 1

    def _capture_step(self, w_text, w_mel, mode, opt):
        """Record the step into hipGraphs: mode "none": (forward + backward + optimizer,); "whole": (forward + backward,
        optimizer) around one collective; "range": (forward + backward part 0, backward part 1, optimizer).
        With a process group alive its watchdog thread polls events while we record, so those cases capture in thread-local
        mode.  A refused capture is reported once on stderr and the engine keeps running launch by launch -- slower on the
        host, identical on the device."""
        torch.cuda.synchronize()
        alive = torch.distributed.is_available() and torch.distributed.is_initialized()
        cmode = "global" if (mode == "none" and not alive) else "thread_local"

        def record(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=cmode):
                fn()
            return g
        try:
            if mode == "none":
                def whole():
                    self.forward(); self.backward(w_text, w_mel); self.optimizer_step(**opt)
                return (record(whole), None)
            if mode == "whole":
                def fb():
                    self.forward(); self.backward(w_text, w_mel)
                return (record(fb), record(lambda: self.optimizer_step(**opt)))
            split = self.grad_exchange_plan()[0]

            def fb0():
                self.forward(); self.backward(w_text, w_mel, part=0, split=split)
            return (record(fb0), record(lambda: self.backward(w_text, w_mel, part=1, split=split)),
                    record(lambda: self.optimizer_step(**opt)))
        except RuntimeError as err:
            import sys
            self._capture_error = "hipGraph capture refused (%s): the step runs launch by launch, collectives eager" % str(err).splitlines()[0][:160]
            print("ttts_amd: " + self._capture_error, file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            self.grads.zero_()     # a partially recorded step leaves nothing behind, but be explicit
            return (None, None)

    def losses(self):
        """(loss_text, loss_mel) as Python floats -- host sync; call it off the hot path."""
        t = self.b["losses"].tolist()
        return t[0], t[1]

    def mel_logits(self):
        """bf16 (B, classes, positions) view, the reference's permuted layout (model.py:441)."""
        B, Tt, Tm = self._bufs_key
        return self.b["logits_m"][:, :self.nm].view(B, Tm, self.nm).permute(0, 2, 1)
