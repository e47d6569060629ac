"""Tensor-level wrappers over the C ABI (include/ttts_hip.h).  PyTorch is plumbing only: device memory,
the current HIP stream and (elsewhere) torch.distributed.  Every function requires tensors on a ROCm device and
raises `TttsError` on any failure -- there is no eager / CPU fallback.
"""
import ctypes

import numpy as np
import os

import torch

from . import lib as _l
from .lib import (EPI_DGELU_BF16, EPI_GELU_BF16, EPI_RESID_ADD_F32, EPI_STORE_BF16, EPI_STORE_F32, TttsError, check)

__all__ = ["gemm_nt", "gemm_tn_accum", "colsum_accum", "attn_fwd", "attn_bwd", "layernorm_fwd", "layernorm_bwd",
           "embed_fwd", "embed_bwd", "ce_fwd", "ce_bwd", "gradnorm", "adamw", "adamw_schedule", "vq_nearest",
           "vq_commit", "vq_ema_update", "stft_mag", "mel_log", "CastPlan", "TransposePlan", "ColsumPlan", "LnFinalizePlan", "TnPlan", "probe_layout", "device_info"]


def _raw_stream(index=None):
    """Handle of torch's current HIP stream on device `index` (default: the current device) as an int.  torch.cuda.current_stream()
    builds a Stream object through three Python layers (~5 us); this is called for every launch -- 5 800 times per VQ-VAE-GAN step,
    whose eager multi-stream form is host-sensitive -- so the raw accessor the runtime itself uses is taken instead."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice() if index is None else index)


def _stream():
    return ctypes.c_void_p(_raw_stream())


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise TttsError("%s must live on the GPU (no CPU fallback in ttts_amd)" % name)
    if t.dtype != dtype:
        raise TttsError("%s must be %s, got %s" % (name, dtype, t.dtype))


def device_info():
    out = (ctypes.c_int32 * 4)()
    check(_l.get().ttts_device_info(out), "device_info")
    return {"arch": out[0], "cus": out[1], "wave": out[2], "lds_per_cu": out[3]}


def _ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise TttsError("expected a 2-D row-major (inner-contiguous) tensor, got strides %s" % (t.stride(),))
    return t.stride(0)


def gemm_nt_plan(M, N, K, epilogue=0):
    """Which kernel / grid `gemm_nt` uses for a shape (host-side query of the library's dispatch table; needs no GPU):
    a `lib.GemmNtPlan` (kernel = lib.NT_KERNEL_*, grid, block, tile_m, tile_n, phase, main_row_tiles, tail_tile_rows)."""
    pl = _l.GemmNtPlan()
    check(_l.get().ttts_gemm_nt_plan_query(int(M), int(N), int(K), int(epilogue), ctypes.byref(pl)), "gemm_nt_plan")
    return pl


def gemm_nt(a, b, c, bias=None, aux=None, epilogue=EPI_STORE_BF16, n=None, k=None, resid_in=None, dropout_p=0.0,
            seed=0, counter=None, colsum=None):
    """c[M,N] = epi(a[M,K] @ b[N,K]^T).  a/b bf16 row-major views; c bf16 or f32 (per epilogue).
    colsum (f32 [>= N], STORE_BF16 / DGELU_BF16): += the column sums of c, taken in the epilogue."""
    _req(resid_in, torch.float32, "resid_in"); _req(colsum, torch.float32, "colsum")
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b"); _req(bias, torch.float32, "bias")
    _req(aux, torch.bfloat16, "aux")
    M = a.shape[0]
    K = a.shape[1] if k is None else k
    N = b.shape[0] if n is None else n
    want = torch.float32 if epilogue in (EPI_RESID_ADD_F32, EPI_STORE_F32) else torch.bfloat16
    _req(c, want, "c")
    if resid_in is not None and (resid_in.shape != c.shape or resid_in.stride() != c.stride()):
        raise TttsError("resid_in must have the layout of c")
    check(_l.get().ttts_gemm_nt_bf16_ex(_p(a), _ld(a), _p(b), _ld(b), _p(c), _ld(c), _p(bias), _p(aux), M, N, K,
                                        epilogue, _p(resid_in), dropout_p, seed, _ctr(counter, c, dropout_p), _p(colsum), _stream()), "gemm_nt")
    return c


def gemm_nt_resid_ln(a, b, x_out, gamma, beta, y, mean, rstd, bias=None, resid_in=None, dropout_p=0.0, seed=0, counter=None,
                     eps=1e-5):
    """x_out[M,512] = resid_in + dropout(bf16(a[M,K] @ b[512,K]^T + bias)); y = LayerNorm(x_out; gamma, beta) (bf16 or f32 by y's
    dtype); mean / rstd [M] -- ONE launch, bit-identical to gemm_nt(RESID_ADD_F32) + layernorm_fwd (ttts_gemm_nt_resid_ln_bf16)."""
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b"); _req(bias, torch.float32, "bias"); _req(resid_in, torch.float32, "resid_in")
    _req(x_out, torch.float32, "x_out"); _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    _req(mean, torch.float32, "mean"); _req(rstd, torch.float32, "rstd")
    if y.dtype not in (torch.bfloat16, torch.float32):
        raise TttsError("gemm_nt_resid_ln: y must be bf16 or f32")
    M, K, N = a.shape[0], a.shape[1], b.shape[0]
    for t, nm in ((x_out, "x_out"), (y, "y"), (resid_in, "resid_in")):
        if t is not None and (tuple(t.shape) != (M, N) or not t.is_contiguous()):
            raise TttsError("gemm_nt_resid_ln: %s must be a contiguous [M, N] tensor" % nm)
    check(_l.get().ttts_gemm_nt_resid_ln_bf16(_p(a), _ld(a), _p(b), _ld(b), _p(bias), _p(resid_in), _p(x_out), M, N, K, dropout_p, seed,
                                              _ctr(counter, x_out, dropout_p), _p(gamma), _p(beta), eps, _p(y), int(y.dtype == torch.bfloat16),
                                              _p(mean), _p(rstd), _stream()), "gemm_nt_resid_ln")
    return x_out, y


def gemm_tn_workspace(mo, no, kr, device):
    n = _l.get().ttts_gemm_tn_workspace_bytes(mo, no, kr)
    return torch.empty(max(n, 16) // 4, dtype=torch.float32, device=device)


def gemm_tn_accum(at, bt, c, mo=None, no=None, workspace=None, kr=None):
    """c[Mo,No] += at[Kr,Mo]^T @ bt[Kr,No]  (fp32; split-K slabs in `workspace`, summed deterministically)."""
    _req(at, torch.bfloat16, "at"); _req(bt, torch.bfloat16, "bt"); _req(c, torch.float32, "c")
    Kr = at.shape[0] if kr is None else kr
    if at.shape[0] < Kr or bt.shape[0] < Kr:
        raise TttsError("gemm_tn: kr exceeds the operands' rows")
    Mo = at.shape[1] if mo is None else mo
    No = bt.shape[1] if no is None else no
    need = _l.get().ttts_gemm_tn_workspace_bytes(Mo, No, Kr)
    if workspace is None:
        workspace = gemm_tn_workspace(Mo, No, Kr, at.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise TttsError("gemm_tn workspace too small (%d < %d bytes)" % (workspace.numel() * workspace.element_size(), need))
    check(_l.get().ttts_gemm_tn_bf16_accum_f32(_p(at), _ld(at), _p(bt), _ld(bt), _p(c), _ld(c), Mo, No, Kr,
                                               _p(workspace), _stream()), "gemm_tn")
    return c


TN_GROUP_MAX = 64   # descriptors per grouped launch (include/ttts_hip.h: ttts_tn_desc_prepare)


class TnPlan:
    """Device-resident descriptor table of a grouped weight-gradient GEMM: c_i[Mo,No] += at_i[Kr,Mo]^T @ bt_i[Kr,No] for
    every entry in ONE launch, each 128x128 output tile reduced over its whole Kr by one workgroup (no slabs).
    entries: list of (at bf16 [Kr, Mo], bt bf16 [Kr, No], c f32 [Mo, No]); Kr % 64 == 0 (zero-padded rows)."""

    def __init__(self, entries, device):
        if not entries:
            raise TttsError("TnPlan: no entries")
        arr = (_l.TnDesc * len(entries))()
        for i, (at, bt, c) in enumerate(entries):
            _req(at, torch.bfloat16, "at"); _req(bt, torch.bfloat16, "bt"); _req(c, torch.float32, "c")
            if at.shape[0] != bt.shape[0] or c.shape[0] != at.shape[1] or c.shape[1] != bt.shape[1]:
                raise TttsError("TnPlan entry %d: shapes at %s bt %s c %s" % (i, tuple(at.shape), tuple(bt.shape), tuple(c.shape)))
            arr[i].At, arr[i].Bt, arr[i].C = at.data_ptr(), bt.data_ptr(), c.data_ptr()
            arr[i].ldat, arr[i].ldbt, arr[i].ldc = _ld(at), _ld(bt), _ld(c)
            arr[i].Mo, arr[i].No, arr[i].Kr = at.shape[1], bt.shape[1], at.shape[0]
        total = ctypes.c_int32(0)
        check(_l.get().ttts_tn_desc_prepare(arr, len(entries), ctypes.byref(total)), "tn_desc_prepare")
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self.table = torch.from_numpy(raw).to(device)
        self.n, self.tiles = len(entries), int(total.value)
        self.tile_counts = [int(_l.get().ttts_tn_desc_tiles(at.shape[1], bt.shape[1])) for at, bt, _ in entries]
        self.flops = float(sum(2.0 * at.shape[0] * at.shape[1] * bt.shape[1] for at, bt, _ in entries))   # algorithmic, per launch
        self.bytes = float(sum(2.0 * at.shape[0] * (at.shape[1] + bt.shape[1]) + 8.0 * at.shape[1] * bt.shape[1] for at, bt, _ in entries))   # operands once, C read + written
        self._keep = entries

    def run(self):
        check(_l.get().ttts_gemm_tn_grouped_bf16_accum_f32(_p(self.table), self.n, self.tiles, _stream()), "gemm_tn_grouped")


def tn_desc_tiles(mo, no):
    return int(_l.get().ttts_tn_desc_tiles(mo, no))


def colsum_accum(x, out, n=None):
    _req(x, torch.bfloat16, "x"); _req(out, torch.float32, "out")
    check(_l.get().ttts_colsum_bf16_accum_f32(_p(x), _ld(x), _p(out), x.shape[0], x.shape[1] if n is None else n,
                                              _stream()), "colsum")
    return out


def attn_fwd(q, k, v, o, lse, B, H, S, dh, qkv_strides, o_strides, scale, dropout_p=0.0, seed=0, counter=None):
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
        _req(t, torch.bfloat16, nm)
    _req(lse, torch.float32, "lse")
    check(_l.get().ttts_attn_causal_fwd_bf16(_p(q), _p(k), _p(v), _p(o), _p(lse), B, H, S, dh, qkv_strides[0],
                                             qkv_strides[1], o_strides[0], o_strides[1], scale, dropout_p, seed,
                                             _ctr(counter, q, dropout_p), _stream()), "attn_fwd")


def attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, workspace, B, H, S, dh, qkv_strides, o_strides, scale, dropout_p=0.0,
             seed=0, counter=None):
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (d_o, "d_o"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _req(t, torch.bfloat16, nm)
    _req(lse, torch.float32, "lse")
    if workspace.numel() * workspace.element_size() < _l.get().ttts_attn_bwd_workspace_bytes(B, H, S):
        raise TttsError("attn_bwd workspace too small")
    check(_l.get().ttts_attn_causal_bwd_bf16(_p(q), _p(k), _p(v), _p(o), _p(d_o), _p(lse), _p(dq), _p(dk), _p(dv),
                                             _p(workspace), B, H, S, dh, qkv_strides[0], qkv_strides[1], o_strides[0],
                                             o_strides[1], scale, dropout_p, seed, _ctr(counter, q, dropout_p), _stream()), "attn_bwd")


def attn_cross_fwd(q, k, v, qmask, kmask, H, scale, fill):
    """Fused fp32 attention on (B, C, T) tensors (include/ttts_hip.h: ttts_attn_cross_fwd_f32) -> (out [B, C, Tq], stats [B, H, Tq, 2])."""
    for t_, nm in ((q, "q"), (k, "k"), (v, "v"), (qmask, "qmask"), (kmask, "kmask")):
        _req(t_, torch.float32, nm)
    B, C, Tq = q.shape
    Tk = k.shape[2]
    out = torch.empty_like(q)
    lse = torch.empty(B, H, Tq, 2, dtype=torch.float32, device=q.device)
    check(_l.get().ttts_attn_cross_fwd_f32(_p(q), _p(k), _p(v), _p(qmask), _p(kmask), _p(out), _p(lse), B, H, C // H, Tq, Tk,
                                           scale, fill, _stream()), "attn_cross_fwd")
    return out, lse


def attn_cross_bwd(q, k, v, qmask, kmask, out, dout, lse, H, scale, fill):
    B, C, Tq = q.shape
    Tk = k.shape[2]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(_l.get().ttts_attn_cross_bwd_workspace_bytes(B, H, Tq) // 4, dtype=torch.float32, device=q.device)
    check(_l.get().ttts_attn_cross_bwd_f32(_p(q), _p(k), _p(v), _p(qmask), _p(kmask), _p(out), _p(dout), _p(lse), _p(dq), _p(dk),
                                           _p(dv), _p(ws), B, H, C // H, Tq, Tk, scale, fill, _stream()), "attn_cross_bwd")
    return dq, dk, dv


def attn_dropout_mask(B, H, S, p, seed, device, counter=None):
    m = torch.empty(B, H, S, S, dtype=torch.uint8, device=device)
    check(_l.get().ttts_attn_dropout_mask_u8(_p(m), B, H, S, p, seed, _ctr(counter, m, p), _stream()), "dropout_mask")
    return m


def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps=1e-5, split=(0, 0)):
    _req(x, torch.float32, "x"); _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    M, D = x.shape
    check(_l.get().ttts_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), int(y.dtype == torch.bfloat16), _p(mean),
                                      _p(rstd), M, D, eps, split[0], split[1], _stream()), "layernorm_fwd")
    return y


def layernorm_bwd_workspace(M, D, device):
    return torch.empty(_l.get().ttts_layernorm_bwd_workspace_bytes(M, D) // 4, dtype=torch.float32, device=device)


def layernorm_bwd(dy, x, gamma, mean, rstd, dx_in, dx, dx_bf16, dgamma, dbeta, workspace, split=(0, 0), dropout_p=0.0,
                  seed=0, dcolsum=None, counter=None):
    _req(x, torch.float32, "x"); _req(dx, torch.float32, "dx"); _req(dx_bf16, torch.bfloat16, "dx_bf16")
    M, D = x.shape
    check(_l.get().ttts_layernorm_bwd_ex(_p(dy), int(dy.dtype == torch.bfloat16), _p(x), _p(gamma), _p(mean), _p(rstd),
                                         _p(dx_in), _p(dx), _p(dx_bf16), _p(dgamma), _p(dbeta), _p(dcolsum),
                                         _p(workspace), M, D, split[0], split[1], dropout_p, seed, _ctr(counter, x, dropout_p),
                                         _stream()), "layernorm_bwd")


def gpt_prepare_tokens(text, mel, mel_valid, Tt, Tm, start_text, stop_text, start_mel, stop_mel, text_inp, text_tar, mel_inp,
                       mel_tar):
    """One launch for UnifiedVoice.forward's token plumbing (include/ttts_hip.h: ttts_gpt_prepare_tokens).
    text / mel: int64 [B, >= Tt / Tm] on the GPU (inner-contiguous); mel_valid: B host ints."""
    for t_, nm in ((text, "text"), (mel, "mel"), (text_inp, "text_inp"), (text_tar, "text_tar"), (mel_inp, "mel_inp"),
                   (mel_tar, "mel_tar")):
        _req(t_, torch.int64, nm)
    B = text.shape[0]
    if mel.shape[0] != B or len(mel_valid) != B:
        raise TttsError("gpt_prepare_tokens: batch sizes differ")
    for t_, n_ in ((text_inp, B * (Tt + 2)), (text_tar, B * (Tt + 2)), (mel_inp, B * (Tm + 2)), (mel_tar, B * (Tm + 2))):
        if t_.numel() != n_ or not t_.is_contiguous():
            raise TttsError("gpt_prepare_tokens: output buffer has %d elements, expected %d (contiguous)" % (t_.numel(), n_))
    valid = (ctypes.c_int32 * B)(*[int(v) for v in mel_valid])
    check(_l.get().ttts_gpt_prepare_tokens(_p(text), _ld(text), _p(mel), _ld(mel), valid, B, Tt, Tm, start_text, stop_text,
                                           start_mel, stop_mel, _p(text_inp), _p(text_tar), _p(mel_inp), _p(mel_tar),
                                           _stream()), "gpt_prepare_tokens")


def embed_fwd(text_inp, mel_inp, text_emb, text_pos, mel_emb, mel_pos, x, dropout_p=0.0, seed=0, counter=None):
    _req(text_inp, torch.int64, "text_inp"); _req(mel_inp, torch.int64, "mel_inp"); _req(x, torch.float32, "x")
    B, Tt = text_inp.shape
    Tm = mel_inp.shape[1]
    D = x.shape[-1]
    check(_l.get().ttts_gpt_embed_fwd(_p(text_inp), _p(mel_inp), _p(text_emb), _p(text_pos), _p(mel_emb), _p(mel_pos),
                                      _p(x), B, Tt, Tm, D, text_emb.shape[0], mel_emb.shape[0], dropout_p, seed,
                                      _ctr(counter, x, dropout_p), _stream()), "embed_fwd")
    return x


def embed_bwd(text_inp, mel_inp, dx, d_text_emb, d_text_pos, d_mel_emb, d_mel_pos, dropout_p=0.0, seed=0, counter=None):
    B, Tt = text_inp.shape
    Tm = mel_inp.shape[1]
    D = dx.shape[-1]
    check(_l.get().ttts_gpt_embed_bwd(_p(text_inp), _p(mel_inp), _p(dx), _p(d_text_emb), _p(d_text_pos), _p(d_mel_emb),
                                      _p(d_mel_pos), B, Tt, Tm, D, dropout_p, seed, _ctr(counter, dx, dropout_p), _stream()),
          "embed_bwd")


def ce_fwd(logits, targets, row_loss, row_lse, loss_mean, C):
    _req(logits, torch.bfloat16, "logits"); _req(targets, torch.int64, "targets")
    check(_l.get().ttts_ce_fwd_bf16(_p(logits), _ld(logits), _p(targets), _p(row_loss), _p(row_lse), _p(loss_mean),
                                    logits.shape[0], C, _stream()), "ce_fwd")


def ce_bwd(logits, targets, row_lse, dlogits, C, grad_scale=1.0, grad_scale_dev=None):
    _req(dlogits, torch.bfloat16, "dlogits")
    check(_l.get().ttts_ce_bwd_bf16(_p(logits), _ld(logits), _p(targets), _p(row_lse), _p(dlogits), grad_scale,
                                    _p(grad_scale_dev), logits.shape[0], C, _stream()), "ce_bwd")


def adamw_schedule(state, base_lr, beta1, beta2, warmup_steps):
    _req(state, torch.float32, "state")
    check(_l.get().ttts_adamw_schedule(_p(state), base_lr, beta1, beta2, warmup_steps, _stream()), "adamw_schedule")


def gradnorm_workspace(n, device):
    return torch.empty(_l.get().ttts_gradnorm_workspace_bytes(n) // 8, dtype=torch.float64, device=device)


def gradnorm(g, max_norm, state, workspace):
    _req(g, torch.float32, "g")
    check(_l.get().ttts_gradnorm_f32(_p(g), g.numel(), max_norm, _p(state), _p(workspace), _stream()), "gradnorm")


def adamw(p, g, m, v, shadow, state, beta1, beta2, eps, wd, zero_grad=True):
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, torch.float32, nm)
    _req(shadow, torch.bfloat16, "shadow")
    check(_l.get().ttts_adamw_f32(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), _p(state), beta1, beta2, eps, wd,
                                  int(zero_grad), _stream()), "adamw")


def _upload(arr, device):
    return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device)


class LnFinalizePlan:
    """Descriptor table for ttts_layernorm_bwd_finalize_batched: entries (workspace, dgamma, dbeta, dcolsum | None), one per
    layernorm_bwd call made with dgamma = dbeta = None (each with its own workspace)."""

    def __init__(self, entries, M, D, device):
        arr = (_l.LnFinalizeDesc * len(entries))()
        for i, (ws, dg, db, dc) in enumerate(entries):
            _req(dg, torch.float32, "dgamma"); _req(db, torch.float32, "dbeta")
            arr[i].workspace, arr[i].dgamma, arr[i].dbeta = ws.data_ptr(), dg.data_ptr(), db.data_ptr()
            arr[i].dcolsum = dc.data_ptr() if dc is not None else None
        self.table, self.n, self.M, self.D, self._keep = _upload(arr, device), len(entries), M, D, entries

    def run(self):
        check(_l.get().ttts_layernorm_bwd_finalize_batched(_p(self.table), self.n, self.M, self.D, _stream()), "ln_finalize_batched")


class WeightNormPlan:
    """Descriptor table for ttts_weight_norm_{fwd,bwd}_batched_f32: entries = dicts of fp32 tensors
    {v [rows, ...], g [rows, ...], w (like v), norm [rows], dw (like v) | None, dv (like v) | None, dg (like g) | None}; one launch
    covers all of them (one workgroup per weight row)."""

    def __init__(self, entries, device):
        if not entries:
            raise TttsError("WeightNormPlan: no entries")
        arr = (_l.WnDesc * len(entries))()
        rows_total = 0
        for i, e in enumerate(entries):
            v = e["v"]
            for k in ("v", "g", "w", "norm", "dw", "dv", "dg"):
                t = e.get(k)
                if t is not None:
                    _req(t, torch.float32, k)
                    if not t.is_contiguous():
                        raise TttsError("WeightNormPlan: %s must be contiguous" % k)
            rows, n = v.shape[0], v.numel() // v.shape[0]
            ptr = lambda k: (e[k].data_ptr() if e.get(k) is not None else None)   # noqa: E731
            arr[i].v, arr[i].g, arr[i].w, arr[i].norm = ptr("v"), ptr("g"), ptr("w"), ptr("norm")
            arr[i].dw, arr[i].dv, arr[i].dg = ptr("dw"), ptr("dv"), ptr("dg")
            arr[i].rows, arr[i].n, arr[i].row_begin = rows, n, rows_total
            rows_total += rows
        self.table, self.n, self.rows, self._keep = _upload(arr, device), len(entries), rows_total, entries

    def forward(self):
        check(_l.get().ttts_weight_norm_fwd_batched_f32(_p(self.table), self.n, self.rows, _stream()), "weight_norm_fwd_batched")

    def backward(self):
        check(_l.get().ttts_weight_norm_bwd_batched_f32(_p(self.table), self.n, self.rows, _stream()), "weight_norm_bwd_batched")


# The weight-split caches and weight-gradient arenas alive on each device.  The C library keeps no list of them (ABI v10): every
# convolution context (one per stream, _conv_ctx) carries the device's handles in `ttts_conv_ctx::handles`; this host-side table is
# where those arrays are rebuilt when an object is created or closed.
_conv_handles = {}            # device ordinal -> {"objs": [owner objects], "array": ctypes array | None}


def _ordinal(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def _check_no_overlap(kind, tensor):
    """A second cache / arena over (part of) the same array is refused BEFORE its storage is allocated: the library (ABI v10) keeps
    no registry, so the check lives with the list the contexts are built from.  Callers that share networks (a second VqvaeStep
    over the same nets) catch the TttsError and run without their own cache, served by the first object's."""
    ent = _conv_handles.get(_ordinal(tensor.device))
    lo, hi = tensor.data_ptr(), tensor.data_ptr() + tensor.numel() * tensor.element_size()
    for o in (ent["objs"] if ent is not None else []):
        if type(o).__name__ != kind or o._h is None:
            continue
        t = o.weights if kind == "WeightSplitCache" else o.grads
        olo, ohi = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
        if lo < ohi and olo < hi:
            raise TttsError("%s: the range overlaps one that already has a %s (close() that one first)" % (kind, kind))


def _register_conv_handle(obj, device):
    ent = _conv_handles.setdefault(_ordinal(device), {"objs": [], "array": None})
    ent["objs"].append(obj)
    _rebuild_conv_handles(_ordinal(device))


def _unregister_conv_handle(obj, device):
    ent = _conv_handles.get(_ordinal(device))
    if ent is not None and obj in ent["objs"]:
        ent["objs"].remove(obj)
        _rebuild_conv_handles(_ordinal(device))


def _rebuild_conv_handles(dkey):
    ent = _conv_handles[dkey]
    hs = [o._h.value for o in ent["objs"] if o._h is not None and o._h.value]
    ent["array"] = (ctypes.c_void_p * len(hs))(*hs) if hs else None
    for (ck, _stream_h), (ctx, _buf) in _conv_ctxs.items():
        if ck == dkey:
            _set_ctx_handles(ctx, ent)


def _set_ctx_handles(ctx, ent):
    arr = ent["array"] if ent is not None else None
    ctx.n_handles = len(arr) if arr is not None else 0
    ctx.handles = ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p)) if arr is not None else ctypes.POINTER(ctypes.c_void_p)()


class WeightSplitCache:
    """Persistent bf16 hi/lo splits of every convolution weight living in `weights` (a flat fp32 tensor: an optimizer's parameter
    arena or a WeightNormBank's effective weights), rewritten by ONE launch per `refresh()` instead of one split launch in front
    of every convolution call (include/ttts_hip.h: ttts_conv_wsplit_cache_*).  `refresh()` after every change of the weights and
    before their next use; `disarm()` when the step ends (calls then split per launch again).  Results never change."""

    def __init__(self, weights, bytes_per_elem=16, slack=64 << 20, max_entries=8192):
        _req(weights, torch.float32, "weights")
        if not weights.is_contiguous():
            raise TttsError("WeightSplitCache: weights must be contiguous")
        _check_no_overlap("WeightSplitCache", weights)
        self.weights = weights
        self.storage = torch.empty(weights.numel() * bytes_per_elem + slack + max_entries * 96, dtype=torch.uint8, device=weights.device)
        h = ctypes.c_void_p()
        check(_l.get().ttts_conv_wsplit_cache_create(_p(weights), weights.numel() * 4, _p(self.storage), self.storage.numel(),
                                                     max_entries, ctypes.byref(h)), "wsplit_cache_create")
        self._h = h
        _register_conv_handle(self, weights.device)

    def refresh(self):
        check(_l.get().ttts_conv_wsplit_cache_refresh(self._h, _stream()), "wsplit_cache_refresh")

    def disarm(self):
        check(_l.get().ttts_conv_wsplit_cache_disarm(self._h), "wsplit_cache_disarm")

    def stats(self):
        out = (ctypes.c_int64 * 4)()
        check(_l.get().ttts_conv_wsplit_cache_stats(self._h, out), "wsplit_cache_stats")
        return {"entries": out[0], "bytes_used": out[1], "hits": out[2], "misses": out[3]}

    def close(self):
        if self._h is not None and self._h.value:
            h, self._h = self._h, None
            _unregister_conv_handle(self, self.weights.device)      # (out of every context before the object goes away)
            _l.get().ttts_conv_wsplit_cache_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WgradSlabArena:
    """Persistent split-K slabs for every convolution weight gradient accumulating into `grads` (a flat fp32 tensor: a
    WeightNormBank's dW buffer or the optimizer's gradient arena), summed by ONE launch in `reduce()` instead of a reduce launch
    behind every weight-gradient call (include/ttts_hip.h: ttts_conv_wgrad_arena_*).  `begin()` before the backward, `reduce()`
    after it and before anything reads the gradients."""

    def __init__(self, grads, storage_bytes, max_entries=8192):
        _req(grads, torch.float32, "grads")
        if not grads.is_contiguous():
            raise TttsError("WgradSlabArena: grads must be contiguous")
        _check_no_overlap("WgradSlabArena", grads)
        self.grads = grads
        self.storage = torch.empty(int(storage_bytes) + max_entries * 64 + 4096, dtype=torch.uint8, device=grads.device)
        h = ctypes.c_void_p()
        check(_l.get().ttts_conv_wgrad_arena_create(_p(grads), grads.numel() * 4, _p(self.storage), self.storage.numel(),
                                                    max_entries, ctypes.byref(h)), "wgrad_arena_create")
        self._h = h
        _register_conv_handle(self, grads.device)

    def begin(self):
        check(_l.get().ttts_conv_wgrad_arena_begin(self._h), "wgrad_arena_begin")

    def reduce(self):
        check(_l.get().ttts_conv_wgrad_arena_reduce(self._h, _stream()), "wgrad_arena_reduce")

    def disarm(self):
        check(_l.get().ttts_conv_wgrad_arena_disarm(self._h), "wgrad_arena_disarm")

    def stats(self):
        out = (ctypes.c_int64 * 6)()
        check(_l.get().ttts_conv_wgrad_arena_stats(self._h, out), "wgrad_arena_stats")
        return {"entries": out[0], "bytes_used": out[1], "deferred": out[2], "fallbacks": out[3], "partial_reduces": out[4],
                "generation": out[5]}

    def release_graphs(self):
        """No hipGraph recorded over this arena will be replayed any more (the caller dropped it): entries frozen by a capture may
        follow new batch shapes again."""
        check(_l.get().ttts_conv_wgrad_arena_release_graphs(self._h), "wgrad_arena_release_graphs")

    def close(self):
        if self._h is not None and self._h.value:
            h, self._h = self._h, None
            _unregister_conv_handle(self, self.grads.device)
            _l.get().ttts_conv_wgrad_arena_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ColsumPlan:
    """Descriptor table for ttts_colsum_bf16_accum_f32_batched: entries (X bf16 [M, ld], out f32 [>= N], N | None)."""

    def __init__(self, entries, device):
        if not 0 < len(entries) <= 64:
            raise TttsError("ColsumPlan: 1 .. 64 problems per launch")
        arr = (_l.ColsumDesc * len(entries))()
        tiles = 0
        for i, (X, out, n) in enumerate(entries):
            _req(X, torch.bfloat16, "X"); _req(out, torch.float32, "out")
            M, N = X.shape[0], (X.shape[1] if n is None else n)
            arr[i].X, arr[i].out, arr[i].ldx, arr[i].M, arr[i].N, arr[i].tile_begin = X.data_ptr(), out.data_ptr(), _ld(X), M, N, tiles
            tiles += _l.get().ttts_colsum_desc_tiles(M, N)
        self.table, self.n, self.tiles, self._keep = _upload(arr, device), len(entries), tiles, entries

    def run(self):
        check(_l.get().ttts_colsum_bf16_accum_f32_batched(_p(self.table), self.n, self.tiles, _stream()), "colsum_batched")


class TransposePlan:
    """Descriptor tables for ttts_transpose_bf16_batched: entries (src bf16 [r, c] contiguous, dst bf16 [c, >= r] row-major);
    more than 64 entries run as several launches."""

    def __init__(self, entries, device):
        self.launches = []
        for c0 in range(0, len(entries), 64):
            chunk = entries[c0:c0 + 64]
            arr = (_l.TransposeDesc * len(chunk))()
            tiles = 0
            for i, (src, dst) in enumerate(chunk):
                _req(src, torch.bfloat16, "transpose src"); _req(dst, torch.bfloat16, "transpose dst")
                r, c = src.shape
                if not src.is_contiguous() or dst.shape[0] != c or dst.stride(0) < r or dst.stride(1) != 1:
                    raise TttsError("transpose: src must be contiguous [r, c], dst a row-major [c, >= r] view")
                arr[i].src, arr[i].dst, arr[i].rows, arr[i].cols, arr[i].ldd, arr[i].tile_begin = \
                    src.data_ptr(), dst.data_ptr(), r, c, dst.stride(0), tiles
                tiles += _l.get().ttts_transpose_desc_tiles(r, c)
            self.launches.append((_upload(arr, device), len(chunk), tiles))
        self._keep = entries

    def run(self):
        for table, n, tiles in self.launches:
            check(_l.get().ttts_transpose_bf16_batched(_p(table), n, tiles, _stream()), "transpose_batched")


class CastPlan:
    """Device-resident descriptor table for the batched fp32 -> bf16 (+ transposed) shadow-weight cast."""

    def __init__(self, entries, device):
        """entries: list of (src f32 [r,c], dst bf16 [r,c] or None, dst_t bf16 [c,r(+pad)] or None)."""
        arr = (_l.CastDesc * len(entries))()
        tiles = 0
        for i, (src, dst, dst_t) in enumerate(entries):
            r, c = src.shape
            _req(src, torch.float32, "cast src")
            if dst_t is not None and (dst_t.shape[0] != c or dst_t.stride(0) < r or dst_t.stride(1) != 1):
                raise TttsError("transposed shadow must be a row-major [cols, >= rows] view")
            arr[i].src, arr[i].dst, arr[i].dst_t = src.data_ptr(), (dst.data_ptr() if dst is not None else None), \
                (dst_t.data_ptr() if dst_t is not None else None)
            arr[i].rows, arr[i].cols, arr[i].tile_begin = r, c, tiles
            arr[i].ldt = dst_t.stride(0) if dst_t is not None else 0
            tiles += _l.get().ttts_cast_desc_tiles(r, c)
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self.table = torch.from_numpy(raw).to(device)
        self.n, self.tiles = len(entries), tiles
        self._keep = entries

    def run(self):
        check(_l.get().ttts_cast_bf16_batched(_p(self.table), self.n, self.tiles, _stream()), "cast_batched")


def vq_nearest(x, codebook, want_xq=True, want_dist=False):
    """x f32 [N,D], codebook f32 [K,D] -> (idx int64 [N], xq f32 [N,D] | None, dist f32 [N] | None)."""
    _req(x, torch.float32, "x"); _req(codebook, torch.float32, "codebook")
    x = x.contiguous(); codebook = codebook.contiguous()
    N, D = x.shape
    K = codebook.shape[0]
    idx = torch.empty(N, dtype=torch.int64, device=x.device)
    xq = torch.empty_like(x) if want_xq else None
    dist = torch.empty(N, dtype=torch.float32, device=x.device) if want_dist else None
    ws = torch.empty(_l.get().ttts_vq_workspace_bytes(N, K), dtype=torch.uint8, device=x.device)
    check(_l.get().ttts_vq_nearest_f32(_p(x), _p(codebook), _p(idx), _p(xq), _p(dist), _p(ws), N, K, D, _stream()),
          "vq_nearest")
    return idx, xq, dist


def vq_commit(x, xq, dx=None, grad_scale=1.0):
    x = x.contiguous(); xq = xq.contiguous()
    N, D = x.shape
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    ws = torch.empty(_l.get().ttts_vq_workspace_bytes(N, 1), dtype=torch.uint8, device=x.device)
    check(_l.get().ttts_vq_commit_f32(_p(x), _p(xq), _p(loss), _p(dx), grad_scale, N, D, _p(ws), _stream()),
          "vq_commit")
    return loss


def vq_ema_update(x, idx, cluster_size, embed_avg, embed, decay=0.99, epsilon=1e-5):
    x = x.contiguous()
    N, D = x.shape
    K = embed.shape[0]
    ws = torch.empty(_l.get().ttts_vq_ema_workspace_bytes(K, D), dtype=torch.uint8, device=x.device)
    check(_l.get().ttts_vq_ema_update_f32(_p(x), _p(idx), _p(cluster_size), _p(embed_avg), _p(embed), _p(ws), N, K, D,
                                          decay, epsilon, _stream()), "vq_ema_update")


_twiddle_cache = {}


def stft_twiddle(n_fft, device):
    key = (n_fft, str(device))
    if key not in _twiddle_cache:
        host = np.empty(n_fft, dtype=np.float32)
        check(_l.get().ttts_stft_twiddle_host(host.ctypes.data_as(ctypes.c_void_p), n_fft), "stft_twiddle")
        _twiddle_cache[key] = torch.from_numpy(host).to(device)
    return _twiddle_cache[key]


def stft_mag(wav, window, n_fft, hop):
    """wav f32 [B,T] -> spec f32 [B, n_fft/2+1, frames] (reflect pad, hann, |.| with 1e-6 floor)."""
    _req(wav, torch.float32, "wav"); _req(window, torch.float32, "window")
    wav = wav.contiguous()
    B, T = wav.shape
    pad = (n_fft - hop) // 2
    frames = (T + 2 * pad - n_fft) // hop + 1
    spec = torch.empty(B, n_fft // 2 + 1, frames, dtype=torch.float32, device=wav.device)
    check(_l.get().ttts_stft_mag_fwd_f32(_p(wav), _p(window), _p(stft_twiddle(n_fft, wav.device)), _p(spec), B, T,
                                         n_fft, hop, _stream()), "stft_mag")
    return spec


# ---- diffusion mel-denoiser pieces (csrc/diffusion_ops.hip) ------------------------------------------------------------------------
def groupnorm_fwd(x, gamma, beta, groups, ss=None, silu=False, eps=1e-5):
    x = _f32c(x, "x")
    B, C, T = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(_l.get().ttts_groupnorm_fwd_f32(_p(x), _p(gamma), _p(beta), _p(ss), _p(y), _p(mean), _p(rstd), B, C, T, groups, eps,
                                          int(silu), _stream()), "groupnorm_fwd")
    return y, mean, rstd


def groupnorm_bwd(dy, x, gamma, beta, ss, mean, rstd, groups, silu=False, dgamma=None, dbeta=None):
    """Returns (dx, dgamma, dbeta, d_scale_shift); dgamma / dbeta given -> accumulated into them in place."""
    dy = _f32c(dy, "dy")
    B, C, T = x.shape
    dx = torch.empty_like(x)
    acc = dgamma is not None
    if not acc:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    dss = torch.empty(B, 2 * C, dtype=torch.float32, device=x.device) if ss is not None else None
    ws = torch.empty(2 * B * C, dtype=torch.float32, device=x.device)
    check(_l.get().ttts_groupnorm_bwd_f32(_p(dy), _p(x), _p(gamma), _p(beta), _p(ss), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta),
                                          _p(dss), _p(ws), B, C, T, groups, int(silu), int(acc), _stream()), "groupnorm_bwd")
    return dx, dgamma, dbeta, dss


def relpos_bias_fwd(table, bucket, H, Tq, Tk, scale):
    _req(table, torch.float32, "table"); _req(bucket, torch.int32, "bucket")
    off = (bucket.numel() - 1) // 2
    bias = torch.empty(H, Tq, Tk, dtype=torch.float32, device=table.device)
    check(_l.get().ttts_relpos_bias_fwd_f32(_p(table), _p(bucket), _p(bias), H, Tq, Tk, off, float(scale), _stream()), "relpos_bias_fwd")
    return bias


def relpos_bias_bwd(dS, bucket, num_buckets, scale, out=None):
    B, H, Tq, Tk = dS.shape
    off = (bucket.numel() - 1) // 2
    acc = out is not None
    if not acc:
        out = torch.empty(num_buckets, H, dtype=torch.float32, device=dS.device)
    ws = torch.empty(_l.get().ttts_relpos_bias_bwd_workspace_bytes(B, H, Tq, Tk) // 4, dtype=torch.float32, device=dS.device)
    check(_l.get().ttts_relpos_bias_bwd_f32(_p(dS), _p(bucket), _p(out), _p(ws), B, H, Tq, Tk, off, num_buckets, float(scale), int(acc),
                                            _stream()), "relpos_bias_bwd")
    return out


def attn_relpos_max_t(products):
    return int(_l.get().ttts_attn_relpos_max_t(int(products)))


def attn_relpos_fwd(qkv, table, bucket, H, scale, products=3):
    """Fused non-causal attention with the bucketed relative-position bias (include/ttts_hip.h: ttts_attn_relpos_fwd_f32).
    qkv (B, 3 H ch, T) fp32 contiguous, per head (q | k | v) x ch channels; returns (out (B, H ch, T), lse (B, H, T))."""
    _req(qkv, torch.float32, "qkv"); _req(table, torch.float32, "table"); _req(bucket, torch.int32, "bucket")
    B, W, T = qkv.shape
    ch = W // (3 * H)
    out = torch.empty(B, H * ch, T, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    check(_l.get().ttts_attn_relpos_fwd_f32(_p(qkv), _p(table), _p(bucket), (bucket.numel() - 1) // 2, _p(out), _p(lse), B, H, T, ch,
                                            float(scale), int(products), _stream()), "attn_relpos_fwd")
    return out, lse


def attn_relpos_bwd(qkv, table, bucket, out, dout, lse, H, scale, products=3, dtable=None, need_dtable=True):
    """dqkv (and the bias table's gradient: added into `dtable` when given -- a gradient-arena slot -- else returned) of
    attn_relpos_fwd.  Returns (dqkv, dtable or None)."""
    B, W, T = qkv.shape
    ch = W // (3 * H)
    dqkv = torch.empty_like(qkv)
    acc = dtable is not None
    if need_dtable and dtable is None:
        dtable = torch.empty_like(table)
    ws = torch.empty(_l.get().ttts_attn_relpos_workspace_bytes(B, H, T) // 4, dtype=torch.float32, device=qkv.device)
    check(_l.get().ttts_attn_relpos_bwd_f32(_p(qkv), _p(table), _p(bucket), (bucket.numel() - 1) // 2, _p(out), _p(dout), _p(lse),
                                            _p(dqkv), _p(dtable) if need_dtable else None, 1 if acc else 0, _p(ws), B, H, T, ch,
                                            table.shape[0], float(scale), int(products), _stream()), "attn_relpos_bwd")
    return dqkv, (dtable if need_dtable else None)


def softmax_bias_fwd(S, bias):
    B, H, Tq, Tk = S.shape
    check(_l.get().ttts_softmax_bias_fwd_f32(_p(S), _p(bias), B, H, Tq, Tk, _stream()), "softmax_bias_fwd")
    return S


def interp_nearest_fwd(x, Tout):
    x = _f32c(x, "x")
    B, C, Tin = x.shape
    y = torch.empty(B, C, Tout, dtype=torch.float32, device=x.device)
    check(_l.get().ttts_interp_nearest_fwd_f32(_p(x), _p(y), B * C, Tin, Tout, _stream()), "interp_nearest_fwd")
    return y


def interp_nearest_bwd(dy, Tin):
    dy = _f32c(dy, "dy")
    B, C, Tout = dy.shape
    dx = torch.empty(B, C, Tin, dtype=torch.float32, device=dy.device)
    check(_l.get().ttts_interp_nearest_bwd_f32(_p(dy), _p(dx), B * C, Tin, Tout, _stream()), "interp_nearest_bwd")
    return dx


_freq_cache = {}


def timestep_embedding(t, dim, max_period=10000.0):
    import math
    _req(t, torch.int64, "t")
    key = (dim, float(max_period), str(t.device))
    if key not in _freq_cache:       # the reference's own fp32 expression (aa_model.py:43-45), evaluated once on the host
        half = dim // 2
        _freq_cache[key] = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(t.device)
    emb = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
    check(_l.get().ttts_timestep_embedding_f32(_p(t), _p(_freq_cache[key]), _p(emb), t.shape[0], dim, _stream()), "timestep_embedding")
    return emb


def select_rows_fwd(use, a, vec):
    _req(use, torch.uint8, "use")
    a = _f32c(a, "a")
    B, C, T = a.shape
    out = torch.empty_like(a)
    check(_l.get().ttts_select_rows_fwd_f32(_p(use), _p(a), _p(vec), _p(out), B, C, T, _stream()), "select_rows_fwd")
    return out


def select_rows_bwd(use, dout, need_da=True, dvec_out=None):
    dout = _f32c(dout, "dout")
    B, C, T = dout.shape
    da = torch.empty_like(dout) if need_da else None
    acc = dvec_out is not None
    dvec = dvec_out if acc else torch.empty(C, dtype=torch.float32, device=dout.device)
    check(_l.get().ttts_select_rows_bwd_f32(_p(use), _p(dout), _p(da), _p(dvec), B, C, T, int(acc), _stream()), "select_rows_bwd")
    return da, dvec


def q_sample(x_start, noise, t, table):
    x_start, noise = _f32c(x_start, "x_start"), _f32c(noise, "noise")
    _req(t, torch.int64, "t"); _req(table, torch.float32, "table")
    xt = torch.empty_like(x_start)
    B = x_start.shape[0]
    check(_l.get().ttts_q_sample_f32(_p(x_start), _p(noise), _p(t), _p(table), _p(xt), B, x_start.numel() // B, _stream()), "q_sample")
    return xt


def diffusion_loss_fwd(model_out, x_start, x_t, noise, t, table):
    """Returns (terms (B, 3) = mse | vb | loss, loss_mean (1,))."""
    model_out = _f32c(model_out, "model_out")
    B, C, T = x_start.shape
    terms = torch.empty(B, 3, dtype=torch.float32, device=x_start.device)
    loss = torch.empty(1, dtype=torch.float32, device=x_start.device)
    ws = torch.empty(_l.get().ttts_diffusion_loss_workspace_bytes(B) // 4, dtype=torch.float32, device=x_start.device)
    check(_l.get().ttts_diffusion_loss_fwd_f32(_p(model_out), _p(x_start), _p(x_t), _p(noise), _p(t), _p(table), _p(terms), _p(loss), _p(ws),
                                               B, C, T, _stream()), "diffusion_loss_fwd")
    return terms, loss


def diffusion_loss_bwd(model_out, x_start, x_t, noise, t, table, gout=None):
    B, C, T = x_start.shape
    d = torch.empty_like(model_out)
    check(_l.get().ttts_diffusion_loss_bwd_f32(_p(model_out), _p(x_start), _p(x_t), _p(noise), _p(t), _p(table), _p(gout), _p(d), B, C, T,
                                               _stream()), "diffusion_loss_bwd")
    return d




# ---- autoregressive decoding (csrc/decode.hip) ------------------------------------------------------------------------
def decode_embed(tokens, emb, pos, ctr, pos_offset, x):
    _req(tokens, torch.int64, "tokens"); _req(emb, torch.float32, "emb"); _req(pos, torch.float32, "pos")
    _req(ctr, torch.int32, "ctr"); _req(x, torch.float32, "x")
    M, D = x.shape
    check(_l.get().ttts_decode_embed_f32(_p(tokens), _p(emb), _p(pos), _p(ctr), int(pos_offset), _p(x), M, D, emb.shape[0],
                                         pos.shape[0], _stream()), "decode_embed")
    return x


def linear_decode(x, w, out, bias=None, epilogue=EPI_STORE_BF16, resid=None, ln1=None, ln2=None, n=None):
    """out[M, N] = epi(LN2(LN1(x)) @ w[N, K]^T + bias) for M <= 16 rows (decode steps).  x bf16, or f32 (+ optional
    (gamma, beta) pairs ln1 / ln2).  Rounding points as gemm_nt."""
    _req(w, torch.bfloat16, "w"); _req(bias, torch.float32, "bias"); _req(resid, torch.float32, "resid")
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_cuda:
        raise TttsError("linear_decode: x must be a bf16 or f32 GPU tensor")
    M, K = x.shape
    N = w.shape[0] if n is None else n
    want = torch.float32 if epilogue in (EPI_RESID_ADD_F32, EPI_STORE_F32) else torch.bfloat16
    _req(out, want, "out")
    if resid is not None and (resid.shape != out.shape or resid.stride() != out.stride()):
        raise TttsError("linear_decode: resid must have the layout of out")
    g1, b1 = ln1 if ln1 is not None else (None, None)
    g2, b2 = ln2 if ln2 is not None else (None, None)
    check(_l.get().ttts_linear_decode_bf16(_p(x), _ld(x), int(x.dtype == torch.float32), _p(g1), _p(b1), _p(g2), _p(b2), _p(w),
                                           _ld(w), _p(bias), _p(out), _ld(out), _p(resid), M, N, K, epilogue, _stream()),
          "linear_decode")
    return out


def kv_cache_fill(qkv, k_cache, v_cache, B, S, H, dh, rep=1):
    """qkv bf16 [B*S, 3*H*dh] -> k_cache / v_cache bf16 [B*rep, H, S_max, dh] rows 0..S-1."""
    _req(qkv, torch.bfloat16, "qkv"); _req(k_cache, torch.bfloat16, "k_cache"); _req(v_cache, torch.bfloat16, "v_cache")
    if qkv.shape != (B * S, 3 * H * dh) or not qkv.is_contiguous():
        raise TttsError("kv_cache_fill: qkv must be contiguous [B*S, 3*H*dh]")
    if k_cache.shape != v_cache.shape or k_cache.shape[:2] != (B * rep, H) or k_cache.shape[3] != dh or not k_cache.is_contiguous():
        raise TttsError("kv_cache_fill: caches must be contiguous [B*rep, H, S_max, dh]")
    check(_l.get().ttts_kv_cache_fill_bf16(_p(qkv), _p(k_cache), _p(v_cache), B, S, H, dh, k_cache.shape[2], rep, _stream()),
          "kv_cache_fill")


def attn_decode(qkv, k_cache, v_cache, ctr, out, scale):
    _req(qkv, torch.bfloat16, "qkv"); _req(k_cache, torch.bfloat16, "k_cache"); _req(v_cache, torch.bfloat16, "v_cache")
    _req(ctr, torch.int32, "ctr"); _req(out, torch.bfloat16, "out")
    M, H, S_max, dh = k_cache.shape
    if qkv.shape != (M, 3 * H * dh) or out.shape != (M, H * dh) or not (qkv.is_contiguous() and out.is_contiguous()):
        raise TttsError("attn_decode: qkv [M, 3*H*dh] / out [M, H*dh] contiguous")
    check(_l.get().ttts_attn_decode_bf16(_p(qkv), _p(k_cache), _p(v_cache), _p(ctr), _p(out), M, H, dh, S_max, float(scale),
                                         _stream()), "attn_decode")
    return out


def sample_logits(logits, ctr, tokens, finished, V, history=None, hist_base=0, out=None, row_div=1, repetition_penalty=1.0,
                  typical_mass=0.0, temperature=1.0, top_k=0, top_p=1.0, do_sample=False, eos_token=-1, pad_token=0, seed=0,
                  probs_out=None, u_out=None):
    _req(logits, torch.float32, "logits"); _req(ctr, torch.int32, "ctr"); _req(tokens, torch.int64, "tokens")
    _req(finished, torch.uint8, "finished"); _req(history, torch.int64, "history"); _req(out, torch.int64, "out")
    _req(probs_out, torch.float32, "probs_out"); _req(u_out, torch.float32, "u_out")
    M = tokens.shape[0]
    check(_l.get().ttts_sample_logits_f32(_p(logits), _ld(logits), row_div, M, V, _p(history),
                                          history.stride(0) if history is not None else 0, hist_base, _p(ctr), _p(tokens), _p(out),
                                          out.stride(0) if out is not None else 0, _p(finished), float(repetition_penalty),
                                          float(typical_mass), float(temperature), int(top_k or 0), float(top_p), int(bool(do_sample)),
                                          int(eos_token), int(pad_token), int(seed) & (2 ** 64 - 1), _p(probs_out), _p(u_out),
                                          _stream()), "sample_logits")
    return tokens


def decode_advance(ctr, finished):
    _req(ctr, torch.int32, "ctr"); _req(finished, torch.uint8, "finished")
    check(_l.get().ttts_decode_advance(_p(ctr), _p(finished), finished.shape[0], _stream()), "decode_advance")


PEQ_PEAK, PEQ_LOW_SHELF, PEQ_HIGH_SHELF = 0, 1, 2


def peq_response(freq, gain, q, kind, n_fft, sample_rate):
    """freq / gain (dB) / q f32 [B, NF], kind i32 [NF] -> complex64 [B, n_fft/2+1] product of the biquad responses."""
    for t, nm in ((freq, "freq"), (gain, "gain"), (q, "q")):
        _req(t, torch.float32, nm)
    _req(kind, torch.int32, "kind")
    B, NF = freq.shape
    if gain.shape != freq.shape or q.shape != freq.shape or kind.numel() != NF:
        raise TttsError("peq_response: freq / gain / q must be [B, NF] and kind [NF]")
    H = torch.empty(B, n_fft // 2 + 1, 2, dtype=torch.float32, device=freq.device)
    check(_l.get().ttts_peq_response_f32(_p(freq.contiguous()), _p(gain.contiguous()), _p(q.contiguous()), _p(kind),
                                         _p(H), B, NF, n_fft, float(sample_rate), _stream()), "peq_response")
    return torch.view_as_complex(H)


def stft_filter_istft(wav, window, n_fft, hop, H=None, clamp=True, peak_normalize=True, eps=1e-7):
    """istft(stft(wav, center=True) * H[..., None]) [.clamp(-1, 1)] [/ peak]: wav f32 [B, T] -> f32 [B, hop * (T // hop)].
    H: complex64 [B, n_fft/2+1] or None (identity)."""
    _req(wav, torch.float32, "wav"); _req(window, torch.float32, "window")
    wav = wav.contiguous()
    B, T = wav.shape
    frames = _l.get().ttts_stft_center_frames(T, hop)
    Hr = None
    if H is not None:
        if H.dtype != torch.complex64 or H.shape != (B, n_fft // 2 + 1):
            raise TttsError("stft_filter_istft: H must be complex64 [B, n_fft/2+1]")
        Hr = torch.view_as_real(H.contiguous())
    fr = torch.empty(B, frames, n_fft, dtype=torch.float32, device=wav.device)
    tw = stft_twiddle(n_fft, wav.device)
    check(_l.get().ttts_stft_filter_frames_f32(_p(wav), _p(window), _p(tw), _p(Hr) if Hr is not None else None, _p(fr), B, T,
                                               n_fft, hop, _stream()), "stft_filter_frames")
    out = torch.empty(B, hop * (frames - 1), dtype=torch.float32, device=wav.device)
    peak = torch.empty(B, dtype=torch.int32, device=wav.device)
    check(_l.get().ttts_istft_ola_f32(_p(fr), _p(window), _p(out), _p(peak), B, frames, n_fft, hop, 1 if clamp else 0,
                                      _stream()), "istft_ola")
    if peak_normalize:
        check(_l.get().ttts_peak_scale_f32(_p(out), _p(peak), B, out.shape[1], eps, _stream()), "peak_scale")
    return out


def mel_bands(basis):
    """int32 [n_mels, 2] = first / last non-zero column of every basis row.  The table is attached to the basis tensor itself
    (one host computation per basis, no global cache: two bases used alternately keep their own tables, and a table can never
    outlive -- or be mistaken for that of -- another tensor at a recycled address)."""
    tab = getattr(basis, "_ttts_bands", None)
    if tab is None or tab[0] != basis._version:
        nz = (basis != 0).cpu()
        n_bins = basis.shape[1]
        cols = torch.arange(n_bins)
        lo = torch.where(nz, cols, torch.full_like(cols, n_bins)).min(dim=1).values
        hi = torch.where(nz, cols, torch.full_like(cols, -1)).max(dim=1).values
        tab = (basis._version, torch.stack([lo, hi], dim=1).to(torch.int32).contiguous().to(basis.device))
        basis._ttts_bands = tab
    return tab[1]


def mel_log(spec, basis):
    _req(spec, torch.float32, "spec"); _req(basis, torch.float32, "basis")
    spec = spec.contiguous(); basis = basis.contiguous()
    B, n_bins, frames = spec.shape
    n_mels = basis.shape[0]
    mel = torch.empty(B, n_mels, frames, dtype=torch.float32, device=spec.device)
    check(_l.get().ttts_mel_log_fwd_f32(_p(spec), _p(basis), _p(mel_bands(basis)), _p(mel), B, n_bins, n_mels, frames, _stream()),
          "mel_log")
    return mel


def conv_out_len(lin, k, stride, pad, dil):
    return (lin + 2 * pad - dil * (k - 1) - 1) // stride + 1


_OUT_ACT = {None: 0, "none": 0, "tanh": 1, "lrelu": 2}
_conv_ctxs = {}


def set_conv_precision(mode):
    """'split_bf16' (default: fp32 products as hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32 accumulation), 'exact'
    (fp32-input MFMA, bit-for-bit an fmaf chain) or 'tf32class' (forward and data gradient: operands rounded to fp16 -- the 11 significant
    bits of the TF32 the reference's cuDNN convolutions run with, ttts/vqvae/train.py:34-36 -- ONE fp16 MFMA product per pair, fp32
    accumulation; weight gradients keep the split form; the data gradient's input needs loss scaling to stay in fp16's range:
    VqvaeStep applies it).  Returns the previous mode."""
    if mode not in ("split_bf16", "exact", "tf32class"):
        raise ValueError("conv precision must be 'split_bf16', 'exact' or 'tf32class'")
    prev = _state["conv_precision"]
    _state["conv_precision"] = mode
    _apply_flags()
    return prev


def conv_precision():
    return _state["conv_precision"]


def set_variant_flags(flags):
    """Kernel-selection switches for experiments (tools/kernel_bench.py, heuristics overrides in tests).  0 = shipped defaults."""
    prev = _state["variant"]
    _state["variant"] = int(flags)
    _apply_flags()
    return prev


_state = {"conv_precision": os.environ.get("TTTS_CONV_PRECISION", "split_bf16"), "variant": int(os.environ.get("TTTS_DEBUG_FLAGS", "0") or 0)}
if _state["conv_precision"] not in ("split_bf16", "exact", "tf32class"):     # a typo must not silently select the default kernels
    raise ValueError("TTTS_CONV_PRECISION must be 'split_bf16', 'exact' or 'tf32class' (got %r)" % _state["conv_precision"])


def conv_f16_events(events, reset=True):
    """events (int32[3], device) += the device's {saturated, flushed-to-zero, reserved = 0} counts of the fp32 -> fp16 operand
    conversions of the 'tf32class' convolution mode since the last reset (include/ttts_hip.h: ttts_conv_f16_events).  No sync."""
    _req(events, torch.int32, "events")
    if events.numel() < 3:
        raise TttsError("conv_f16_events: events must hold three int32 words")
    check(_l.get().ttts_conv_f16_events(_p(events), 1 if reset else 0, _stream()), "conv_f16_events")
    return events


class DynamicLossScale:
    """`torch.cuda.amp.GradScaler`'s rule (ttts/vqvae/train.py:262,356-372) for the 'tf32class' convolutions, entirely on the device
    -- no `.item()`, so the step stays one stream of launches and can be recorded as a hipGraph:

        loss * s.scale                   (device scalar)                                   scaler.scale(loss)
        s.check(optimizer[, reduce])     after the backward: range events -> skip flag     scaler.unscale_ / found_inf
        arena.mul_(s.inv_scale)          exact for powers of two                           scaler.unscale_
        optimizer.step()                 a no-op when the check set its skip flag          scaler.step
        s.update()                       halve on overflow, double after `interval` clean  scaler.update

    What counts as overflow: an operand of an fp16 conversion above 65504 (the conversions saturate instead of producing inf, so
    the usual "gradient is inf / nan" test would never fire -- the device counters of ops.conv_f16_events are the test).  State,
    as device scalars: .scale, .inv_scale, .saturated, .flushed, .skipped (totals since construction)."""

    def __init__(self, device, init_scale=1024.0, growth_interval=2000, backoff=0.5, growth=2.0, dynamic=True):
        self.state = torch.zeros(8, dtype=torch.float32, device=device)
        self.state[0] = float(init_scale)
        self.state[1] = 1.0 / float(init_scale)
        self.events = torch.zeros(4, dtype=torch.int32, device=device)
        self.interval, self.backoff, self.growth, self.dynamic = int(growth_interval), float(backoff), float(growth), bool(dynamic)

    @classmethod
    def from_env(cls, device):
        """TTTS_LOSS_SCALE = initial scale (default 2^10), TTTS_LOSS_SCALE_DYNAMIC=0 pins it, TTTS_LOSS_SCALE_INTERVAL = clean
        steps in a row before it doubles (default 2000, GradScaler's)."""
        return cls(device, float(os.environ.get("TTTS_LOSS_SCALE", "1024")), int(os.environ.get("TTTS_LOSS_SCALE_INTERVAL", "2000")),
                   dynamic=os.environ.get("TTTS_LOSS_SCALE_DYNAMIC", "1") != "0")

    scale = property(lambda self: self.state[0])
    inv_scale = property(lambda self: self.state[1])
    saturated = property(lambda self: self.state[4])
    flushed = property(lambda self: self.state[5])
    skipped = property(lambda self: self.state[6])

    def fetch(self):
        """Move the device's range-event counters of the backward that just ran into .events (and clear them)."""
        conv_f16_events(self.events, reset=True)
        return self.events

    def decide(self, optimizer_state=None):
        """An overflow in .events sets optimizer_state[6] (FlatAdamW / GptEngine `opt_state`), which turns the following optimizer
        step into a no-op; totals accumulate; .events is cleared.  A static scale (dynamic=False) counts but never skips."""
        skip = optimizer_state[6:7] if (optimizer_state is not None and self.dynamic) else None
        check(_l.get().ttts_loss_scale_check(_p(self.state), _p(self.events), _p(skip), _stream()), "loss_scale_check")

    def check(self, optimizer_state=None, reduce=None):
        """fetch + decide; `reduce(events)` in between (data parallel: a SUM all-reduce) makes the decision the same on every rank."""
        self.fetch()
        if reduce is not None:
            reduce(self.events)
        self.decide(optimizer_state)

    def update(self):
        if self.dynamic:
            check(_l.get().ttts_loss_scale_update(_p(self.state), self.interval, self.backoff, self.growth, _stream()), "loss_scale_update")
        else:
            self.state[3].zero_()

    def report(self):
        """Host copy (one sync): {'scale', 'saturated', 'flushed', 'skipped'}."""
        v = self.state.tolist()
        return {"scale": v[0], "saturated": int(v[4]), "flushed": int(v[5]), "skipped": int(v[6])}


def _apply_flags():
    for ctx, _buf in _conv_ctxs.values():
        ctx.flags = _state["variant"] | {"exact": 4096, "tf32class": 1024}.get(_state["conv_precision"], 0)


def _conv_ctx(device):
    """The convolution context of `device` (include/ttts_hip.h: ttts_conv_ctx), created on first use: a caller-owned struct
    + the scratch tensor that enables the split-bf16 matrix-core path.  Python-side state only -- the C library keeps none;
    every conv entry point receives the struct by pointer.  TTTS_CONV_FP32=1 creates it without scratch (exact kernels)."""
    # one scratch per (device, stream): convolutions issued on different streams (the sub-discriminators run concurrently,
    # vq2.MultiPeriodDiscriminator) must not share operand-split buffers
    # (keyed by the device ORDINAL and the stream handle: this runs once per convolution call, ~1700 times a step -- no string work here)
    if device.type != "cuda":
        raise TttsError("the convolution family runs on a GPU only (got %s)" % device)
    idx = device.index if device.index is not None else torch._C._cuda_getDevice()
    key = (idx, _raw_stream(idx))
    if key not in _conv_ctxs:
        buf = None
        if os.environ.get("TTTS_CONV_FP32", "0") != "1":
            buf = torch.empty(int(os.environ.get("TTTS_CONV_SCRATCH_MB", "1536")) << 20, dtype=torch.uint8, device=device)
        ctx = _l.ConvCtx(_p(buf), buf.numel() if buf is not None else 0, 0, 0, ctypes.POINTER(ctypes.c_void_p)(), idx, 0)
        _set_ctx_handles(ctx, _conv_handles.get(idx))
        _conv_ctxs[key] = (ctx, buf)
        _apply_flags()
    return ctypes.byref(_conv_ctxs[key][0])



def release_conv_ctxs(device=None, keep_current=True):
    """Drop the per-stream convolution contexts (and their 1.5 GB scratch buffers each) of `device` -- all devices when None -- after a
    device synchronisation; with keep_current the context of torch's current stream stays.  Contexts are otherwise kept for the life of
    the process (one per stream that ever ran a convolution: ~12 for the VQ-VAE-GAN step with its branch streams); call this when a
    set of side streams is retired, e.g. between a training run and an evaluation that uses none.  Returns the number dropped."""
    if not _conv_ctxs:
        return 0
    torch.cuda.synchronize()
    want = None if device is None else _ordinal(device)
    dropped = 0
    for key in list(_conv_ctxs):
        idx, stream = key
        if want is not None and idx != want:
            continue
        if keep_current and stream == _raw_stream(idx):
            continue
        del _conv_ctxs[key]
        dropped += 1
    return dropped


def conv1d_fwd(x, w, bias=None, resid=None, stride=1, pad=0, dil=1, in_slope=1.0, out_act=None, out_scale=1.0,
               out=None, accumulate=False, bbias=None, gate=None, gate_slope=1.0, groups=1, out_slope=1.0, omask=None):
    """y = [y +] out_scale * act(lrelu'(gate) * (bias + bbias + conv1d(lrelu(x, in_slope), w)) + resid);
    x (B,Cin,L) fp32, w (Cout,Cin/groups,K), bbias (B,Cout); out_act None | 'tanh' | 'lrelu' (slope out_slope).
    Also the data gradient of a ConvTranspose1d (w = its weight)."""
    for t, n in ((x, "x"), (w, "w"), (bias, "bias"), (resid, "resid"), (bbias, "bbias"), (gate, "gate")):
        _req(t, torch.float32, n)
    x = x.contiguous(); w = w.contiguous()
    B, Cin, Lin = x.shape
    Cout, _, K = w.shape
    Lout = conv_out_len(Lin, K, stride, pad, dil)
    y = out if out is not None else torch.empty(B, Cout, Lout, dtype=torch.float32, device=x.device)
    c = lambda t: t.contiguous() if t is not None else None
    check(_l.get().ttts_conv1d_fwd_f32(_p(x), _p(w), _p(c(bias)), _p(c(bbias)), _p(c(resid)), _p(c(gate)), _p(c(omask)), _p(y), B, Cin, Lin,
                                       Cout, Lout, K, stride, pad, dil, groups, in_slope, gate_slope, _OUT_ACT[out_act],
                                       out_slope, out_scale, int(accumulate), _conv_ctx(x.device), _stream()), "conv1d_fwd")
    return y


def conv1d_fwd_dual(x, w, bias, resid, omask, y, y2, cout1, pad=0, dil=1, in_slope=1.0, accumulate2=False):
    """One stride-1 convolution with two destinations (include/ttts_hip.h: ttts_conv1d_fwd_dual_f32): output channels [0, cout1) ->
    y = (conv + bias + resid) * omask, channels [cout1, Cout) -> y2 [+]= (conv + bias) * omask.  Returns (y, y2)."""
    for t, n in ((x, "x"), (w, "w"), (bias, "bias"), (resid, "resid"), (omask, "omask"), (y, "y"), (y2, "y2")):
        _req(t, torch.float32, n)
    x = x.contiguous(); w = w.contiguous()
    B, Cin, Lin = x.shape
    Cout, _, K = w.shape
    Lout = conv_out_len(Lin, K, 1, pad, dil)
    if not (y.is_contiguous() and y2.is_contiguous() and tuple(y.shape) == (B, cout1, Lout) and tuple(y2.shape) == (B, Cout - cout1, Lout)):
        raise TttsError("conv1d_fwd_dual: y / y2 must be contiguous (B, cout1, Lout) / (B, Cout - cout1, Lout)")
    c = lambda t: t.contiguous() if t is not None else None
    check(_l.get().ttts_conv1d_fwd_dual_f32(_p(x), _p(w), _p(c(bias)), _p(c(resid)), _p(c(omask)), _p(y), _p(y2), B, Cin, Lin, Cout, cout1,
                                            Lout, K, pad, dil, in_slope, int(accumulate2), _conv_ctx(x.device), _stream()), "conv1d_fwd_dual")
    return y, y2


def conv1d_dgrad(dy, w, lin, stride=1, pad=0, dil=1, gate=None, gate_slope=1.0, bias=None, in_slope=1.0, resid=None,
                 out_scale=1.0, out=None, accumulate=False, groups=1, omask=None):
    """dx (B,Cin,lin) of a conv with weight w (Cout,Cin/groups,K) -- or the ConvTranspose1d forward with w = its weight
    (then `in_slope` is the leaky-relu fused on its input and `bias` its bias)."""
    _req(dy, torch.float32, "dy"); _req(w, torch.float32, "w")
    dy = dy.contiguous(); w = w.contiguous()
    B, Cout, Lout = dy.shape
    Cin, K = w.shape[1] * groups, w.shape[2]
    dx = out if out is not None else torch.empty(B, Cin, lin, dtype=torch.float32, device=dy.device)
    c = lambda t: t.contiguous() if t is not None else None
    check(_l.get().ttts_conv1d_dgrad_f32(_p(dy), _p(w), _p(c(bias)), _p(c(resid)), _p(c(gate)), _p(c(omask)), _p(dx), B, Cin, lin, Cout, Lout,
                                         K, stride, pad, dil, groups, in_slope, gate_slope, out_scale, int(accumulate),
                                         _conv_ctx(dy.device), _stream()), "conv1d_dgrad")
    return dx


def conv1d_wgrad(dy, x, k, stride=1, pad=0, dil=1, x_slope=1.0, dy_slope=1.0, out=None, groups=1, db=None):
    """dw (Cout,Cin/groups,k) [+]= sum_{b,l} lrelu(dy, dy_slope) * lrelu(x, x_slope) (shifted); `out` accumulates.
    db (Cout,), optional: db += sum_{b,l} dy in the same call (the layer's bias gradient; needs dy_slope == 1)."""
    dy = dy.contiguous(); x = x.contiguous()
    B, Cout, Lout = dy.shape
    _, Cin, Lin = x.shape
    dw = out if out is not None else torch.zeros(Cout, Cin // groups, k, dtype=torch.float32, device=dy.device)
    check(_l.get().ttts_conv1d_wgrad_f32(_p(dy), _p(x), _p(dw), _p(db), B, Cin, Lin, Cout, Lout, k, stride, pad, dil, groups, dy_slope,
                                         x_slope, _conv_ctx(dy.device), _stream()), "conv1d_wgrad")
    return dw


def conv1d_bias_grad(dy, out=None):
    """db[c] (+)= sum_{b,l} dy; `out` accumulates in place."""
    dy = dy.contiguous()
    B, C, L = dy.shape
    db = torch.zeros(C, dtype=torch.float32, device=dy.device) if out is None else out
    check(_l.get().ttts_conv1d_bias_grad_f32(_p(dy), _p(db), B, C, L, _stream()), "conv1d_bias_grad")
    return db


def weight_norm_fwd(v, g, want_zero=False):
    """w = g v / |v| per output row.  want_zero: also returns a cleared buffer of w's shape, written by the same kernel (the
    accumulate-into buffer of the layer's weight gradient)."""
    v = v.contiguous()
    rows, n = v.shape[0], v.numel() // v.shape[0]
    w = torch.empty_like(v)
    zero = torch.empty_like(v) if want_zero else None
    norm = torch.empty(rows, dtype=torch.float32, device=v.device)
    check(_l.get().ttts_weight_norm_fwd_f32(_p(v), _p(g.contiguous()), _p(w), _p(norm), _p(zero), rows, n, _stream()), "weight_norm_fwd")
    return (w, norm, zero) if want_zero else (w, norm)


def weight_norm_bwd(dw, v, g, norm, dv=None, dg=None):
    """dv, dg given: accumulate into them (the kernels add), e.g. straight into the flat gradient arena."""
    v = v.contiguous(); dw = dw.contiguous()
    rows, n = v.shape[0], v.numel() // v.shape[0]
    dv = torch.zeros_like(v) if dv is None else dv
    dg = torch.zeros_like(g).contiguous() if dg is None else dg
    check(_l.get().ttts_weight_norm_bwd_f32(_p(dw), _p(v), _p(g.contiguous()), _p(norm), _p(dv), _p(dg), rows, n, _stream()),
          "weight_norm_bwd")
    return dv, dg


def tanh_bwd(dy, y):
    dy = dy.contiguous(); y = y.contiguous()
    dx = torch.empty_like(dy)
    check(_l.get().ttts_tanh_bwd_f32(_p(dy), _p(y), _p(dx), dy.numel(), _stream()), "tanh_bwd")
    return dx


def lrelu_bwd(dy, y, slope):
    dy = dy.contiguous(); y = y.contiguous()
    dx = torch.empty_like(dy)
    check(_l.get().ttts_lrelu_bwd_f32(_p(dy), _p(y), _p(dx), slope, dy.numel(), _stream()), "lrelu_bwd")
    return dx


def add_scale(tensors, scale=1.0):
    """scale * sum(tensors) for 1..4 same-shape fp32 tensors."""
    ts = [t.contiguous() for t in tensors]
    assert 1 <= len(ts) <= 4
    for t in ts:
        _req(t, torch.float32, "add_scale input")
    y = torch.empty_like(ts[0])
    ptrs = [_p(t) for t in ts] + [None] * (4 - len(ts))
    check(_l.get().ttts_add4_scale_f32(*ptrs, scale, _p(y), y.numel(), _stream()), "add4_scale")
    return y


GATE_TANH_SIGMOID, GATE_GLU = 0, 1
ACT_RELU, ACT_MISH, ACT_SILU = 0, 1, 2


def _f32c(t, name):
    _req(t, torch.float32, name)
    return t.contiguous()


def gate_fwd(x, kind=GATE_TANH_SIGMOID):
    x = _f32c(x, "x")
    B, C2, T = x.shape
    y = torch.empty(B, C2 // 2, T, dtype=torch.float32, device=x.device)
    check(_l.get().ttts_gate_fwd_f32(_p(x), _p(y), B, C2 // 2, T, kind, _stream()), "gate_fwd")
    return y


def gate_bwd(dy, x, kind=GATE_TANH_SIGMOID, rowsum=None):
    """rowsum (f32, B * 2H elements, contiguous): also receives sum_t dx[b][c][t] (written, not accumulated)."""
    dy = _f32c(dy, "dy"); x = _f32c(x, "x")
    B, C2, T = x.shape
    dx = torch.empty_like(x)
    if rowsum is not None:
        assert rowsum.is_contiguous() and rowsum.numel() == B * C2 and rowsum.dtype == torch.float32
        check(_l.get().ttts_gate_bwd_rowsum_f32(_p(dy), _p(x), _p(dx), _p(rowsum), B, C2 // 2, T, kind, _stream()), "gate_bwd_rowsum")
        return dx
    check(_l.get().ttts_gate_bwd_f32(_p(dy), _p(x), _p(dx), B, C2 // 2, T, kind, _stream()), "gate_bwd")
    return dx


def mul_mask(x, mask):
    """x (B,C,T) * mask (B,1,T) or (B,T)."""
    x = _f32c(x, "x"); mask = _f32c(mask, "mask")
    B, C, T = x.shape
    assert mask.numel() == B * T
    y = torch.empty_like(x)
    check(_l.get().ttts_mul_mask_f32(_p(x), _p(mask), _p(y), B, C, T, _stream()), "mul_mask")
    return y


def gauss_sample_fwd(stats, eps, mask):
    stats = _f32c(stats, "stats"); eps = _f32c(eps, "eps")
    mask = _f32c(mask, "mask") if mask is not None else None
    B, C2, T = stats.shape
    z = torch.empty(B, C2 // 2, T, dtype=torch.float32, device=stats.device)
    check(_l.get().ttts_gauss_sample_fwd_f32(_p(stats), _p(eps), _p(mask), _p(z), B, C2 // 2, T, _stream()), "gauss_sample_fwd")
    return z


def gauss_sample_bwd(dz, stats, eps, mask):
    dz = _f32c(dz, "dz")
    B, C2, T = stats.shape
    dstats = torch.empty_like(stats)
    check(_l.get().ttts_gauss_sample_bwd_f32(_p(dz), _p(stats), _p(eps), _p(mask), _p(dstats), B, C2 // 2, T, 0, _stream()),
          "gauss_sample_bwd")
    return dstats


def upsample2_fwd(x):
    x = _f32c(x, "x")
    y = torch.empty(*x.shape[:-1], x.shape[-1] * 2, dtype=torch.float32, device=x.device)
    check(_l.get().ttts_upsample2_fwd_f32(_p(x), _p(y), x.numel(), _stream()), "upsample2_fwd")
    return y


def upsample2_bwd(dy):
    dy = _f32c(dy, "dy")
    dx = torch.empty(*dy.shape[:-1], dy.shape[-1] // 2, dtype=torch.float32, device=dy.device)
    check(_l.get().ttts_upsample2_bwd_f32(_p(dy), _p(dx), dx.numel(), _stream()), "upsample2_bwd")
    return dx


def act_fwd(x, op):
    x = _f32c(x, "x")
    y = torch.empty_like(x)
    check(_l.get().ttts_act_fwd_f32(_p(x), _p(y), x.numel(), op, _stream()), "act_fwd")
    return y


def act_bwd(dy, x, op):
    dy = _f32c(dy, "dy")
    dx = torch.empty_like(x)
    check(_l.get().ttts_act_bwd_f32(_p(dy), _p(x), _p(dx), x.numel(), op, _stream()), "act_bwd")
    return dx


def dropout(x, p, seed, counter=None):
    x = _f32c(x, "x")
    y = torch.empty_like(x)
    check(_l.get().ttts_dropout_f32(_p(x), _p(y), x.numel(), p, seed, _ctr(counter, x, p), _stream()), "dropout")
    return y


def snake_aa_fwd(x, alpha, beta, fup, fdn):
    x = _f32c(x, "x")
    B, C, T = x.shape
    y = torch.empty_like(x)
    check(_l.get().ttts_snake_aa_fwd_f32(_p(x), _p(_f32c(alpha, "alpha")), _p(_f32c(beta, "beta")), _p(_f32c(fup, "fup")),
                                         _p(_f32c(fdn, "fdn")), _p(y), B, C, T, _stream()), "snake_aa_fwd")
    return y


def snake_aa_bwd(dy, x, alpha, beta, fup, fdn):
    dy = _f32c(dy, "dy")
    B, C, T = x.shape
    dx = torch.empty_like(x)
    da = torch.zeros_like(alpha); db = torch.zeros_like(beta)
    check(_l.get().ttts_snake_aa_bwd_f32(_p(dy), _p(x), _p(alpha.contiguous()), _p(beta.contiguous()), _p(fup.contiguous()),
                                         _p(fdn.contiguous()), _p(dx), _p(da), _p(db), B, C, T, _stream()), "snake_aa_bwd")
    return dx, da, db


def layernorm_ch_fwd(x, gamma, beta, eps=1e-5):
    x = _f32c(x, "x")
    B, C, T = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(B * T, dtype=torch.float32, device=x.device); rstd = torch.empty_like(mean)
    check(_l.get().ttts_layernorm_ch_fwd_f32(_p(x), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")), _p(y), _p(mean), _p(rstd),
                                             B, C, T, eps, _stream()), "layernorm_ch_fwd")
    return y, mean, rstd


def layernorm_ch_bwd(dy, x, gamma, mean, rstd, dg=None, db=None):
    """dg / db given: the kernels ACCUMULATE into them (e.g. the parameters' .grad slices of a flat gradient arena)."""
    dy = _f32c(dy, "dy")
    B, C, T = x.shape
    dx = torch.empty_like(x)
    dg = torch.zeros_like(gamma) if dg is None else dg
    db = torch.zeros_like(gamma) if db is None else db
    check(_l.get().ttts_layernorm_ch_bwd_f32(_p(dy), _p(x), _p(gamma.contiguous()), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db),
                                             B, C, T, _stream()), "layernorm_ch_bwd")
    return dx, dg, db


def embedding_ct_fwd(idx, table):
    _req(idx, torch.int64, "idx"); table = _f32c(table, "table")
    idx = idx.contiguous()
    B, T = idx.shape
    C = table.shape[1]
    y = torch.empty(B, C, T, dtype=torch.float32, device=table.device)
    check(_l.get().ttts_embedding_ct_fwd_f32(_p(idx), _p(table), _p(y), B, C, T, _stream()), "embedding_ct_fwd")
    return y


def embedding_ct_bwd(idx, dy, rows):
    dy = _f32c(dy, "dy")
    B, C, T = dy.shape
    dt = torch.zeros(rows, C, dtype=torch.float32, device=dy.device)
    check(_l.get().ttts_embedding_ct_bwd_f32(_p(idx.contiguous()), _p(dy), _p(dt), B, C, T, _stream()), "embedding_ct_bwd")
    return dt


def masked_mean_fwd(x, mask):
    x = _f32c(x, "x")
    B, C, T = x.shape
    y = torch.empty(B, C, dtype=torch.float32, device=x.device)
    check(_l.get().ttts_masked_mean_fwd_f32(_p(x), _p(mask), _p(y), B, C, T, _stream()), "masked_mean_fwd")
    return y


def masked_mean_bwd(dy, mask, T):
    dy = _f32c(dy, "dy")
    B, C = dy.shape
    dx = torch.empty(B, C, T, dtype=torch.float32, device=dy.device)
    check(_l.get().ttts_masked_mean_bwd_f32(_p(dy), _p(mask), _p(dx), B, C, T, _stream()), "masked_mean_bwd")
    return dx


def bgemm(A, B, C, M, N, K, a_s, b_s, c_s, outer, inner, a_b, b_b, c_b, alpha=1.0, beta=0.0):
    """C[z][m][n] = alpha sum_k A[z][m][k] B[z][k][n] + beta C; a_s = (s_m, s_k), b_s = (s_k, s_n), c_s = (s_m, s_n),
    *_b = (outer batch stride, inner batch stride), all in elements; tensors are only used for their base pointers."""
    check(_l.get().ttts_bgemm_f32(_p(A), _p(B), _p(C), M, N, K, a_s[0], a_s[1], b_s[0], b_s[1], c_s[0], c_s[1], outer, inner,
                                  a_b[0], a_b[1], b_b[0], b_b[1], c_b[0], c_b[1], alpha, beta, _stream()), "bgemm")
    return C


def attn_softmax_fwd(S, q, ek, qmask, kmask, dk, window, scale, fill):
    B, H, Tq, Tk = S.shape
    hrel = ek.shape[0] if ek is not None else 1
    check(_l.get().ttts_attn_softmax_fwd_f32(_p(S), _p(q), _p(ek), _p(qmask), _p(kmask), B, H, Tq, Tk, dk, window, hrel, scale,
                                             fill, _stream()), "attn_softmax_fwd")
    return S


def attn_softmax_bwd(dP, P, qmask, kmask):
    B, H, Tq, Tk = P.shape
    check(_l.get().ttts_attn_softmax_bwd_f32(_p(dP), _p(P), _p(qmask), _p(kmask), B, H, Tq, Tk, _stream()), "attn_softmax_bwd")
    return dP


def attn_rel(W, X, E, H, window, scale, mode):
    B, C, T = X.shape
    check(_l.get().ttts_attn_rel_f32(_p(W), _p(X), _p(E), B, H, T, C // H, window, E.shape[0], scale, mode, _stream()), "attn_rel")


RED_ABSDIFF, RED_SQ_ONE_MINUS, RED_SQ = 0, 1, 2
_loss_ws = {}


def _loss_workspace(device):
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    if key not in _loss_ws:
        _loss_ws[key] = torch.empty(_l.get().ttts_loss_workspace_bytes(), dtype=torch.uint8, device=device)
    return _loss_ws[key]


def reduce_loss(a, b, mode, scale, out=None, accumulate=False):
    """out[0] = [out[0] +] scale * sum term(a, b) (device scalar, fp32 [1]); a, b contiguous fp32."""
    _req(a, torch.float32, "a"); _req(b, torch.float32, "b")
    assert a.is_contiguous() and (b is None or (b.is_contiguous() and b.numel() == a.numel()))
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(_l.get().ttts_reduce_loss_f32(_p(a), _p(b), a.numel(), mode, scale, _p(out), int(accumulate),
                                        _p(_loss_workspace(a.device)), _stream()), "reduce_loss")
    return out


def reduce_loss_bwd(a, b, mode, scale, gout, out=None, accumulate=False):
    d = out if out is not None else torch.empty_like(a)
    check(_l.get().ttts_reduce_loss_bwd_f32(_p(a), _p(b), a.numel(), mode, scale, _p(gout), _p(d), int(accumulate), _stream()),
          "reduce_loss_bwd")
    return d


def kl_loss_fwd(z_p, logs_q, m_p, logs_p, mask):
    ts = [t.contiguous() for t in (z_p, logs_q, m_p, logs_p, mask)]
    for t in ts:
        _req(t, torch.float32, "kl_loss input")
    B, C, T = ts[0].shape
    out = torch.empty(2, dtype=torch.float32, device=ts[0].device)
    check(_l.get().ttts_kl_loss_fwd_f32(*[_p(t) for t in ts], B, C, T, _p(out), _p(_loss_workspace(out.device)), _stream()),
          "kl_loss_fwd")
    return out


def kl_loss_bwd(z_p, logs_q, m_p, logs_p, mask, out, gout):
    ts = [t.contiguous() for t in (z_p, logs_q, m_p, logs_p, mask)]
    B, C, T = ts[0].shape
    ds = [torch.empty_like(ts[0]) for _ in range(4)]
    check(_l.get().ttts_kl_loss_bwd_f32(*[_p(t) for t in ts], _p(out), _p(gout), B, C, T, *[_p(d) for d in ds], _stream()),
          "kl_loss_bwd")
    return ds


_dropout_counters = {}


def dropout_counter(device):
    """The dropout stream counter of `device` (int32 [1], created on first use, owned here -- the C library is stateless and
    receives the pointer as an explicit argument of every dropout-capable call).  Increment it once per training step
    (`counter.add_(1)`, capturable) for fresh masks; operators called with dropout_p > 0 and no explicit `counter=` use it."""
    key = _device_key(device)
    if key not in _dropout_counters:
        _dropout_counters[key] = torch.zeros(1, dtype=torch.int32, device=key)
    return _dropout_counters[key]


def _device_key(device):
    """'cuda:<index>' for any spelling of a GPU device ('cuda' names the current one): a counter registered under 'cuda' must be
    found by tensors that report 'cuda:0' -- otherwise graph replays would reuse one captured dropout mask."""
    d = torch.device(device)
    if d.type != "cuda":
        return str(d)
    return "cuda:%d" % (d.index if d.index is not None else torch.cuda.current_device())


def _ctr(counter, like, dropout_p):
    """Device pointer of the dropout counter an operator call uses: the explicit one, else (when dropout is on) the
    device's default counter if one was created, else NULL."""
    if counter is None and dropout_p > 0.0:
        counter = _dropout_counters.get(_device_key(like.device))
    return _p(counter)


def stft_mag_bwd(wav, window, dspec, n_fft, hop):
    """Gradient of stft_mag w.r.t. wav: f32 [B, T]."""
    _req(wav, torch.float32, "wav"); _req(dspec, torch.float32, "dspec")
    wav = wav.contiguous(); dspec = dspec.contiguous()
    B, T = wav.shape
    dwav = torch.zeros_like(wav)
    check(_l.get().ttts_stft_mag_bwd_f32(_p(wav), _p(window), _p(stft_twiddle(2 * n_fft, wav.device)), _p(dspec), _p(dwav),
                                         B, T, n_fft, hop, _stream()), "stft_mag_bwd")
    return dwav


def mel_log_bwd(dmel, mel, basis, n_bins):
    _req(dmel, torch.float32, "dmel"); _req(mel, torch.float32, "mel")
    dmel = dmel.contiguous(); mel = mel.contiguous(); basis = basis.contiguous()
    B, n_mels, frames = mel.shape
    dspec = torch.zeros(B, n_bins, frames, dtype=torch.float32, device=mel.device)
    check(_l.get().ttts_mel_log_bwd_f32(_p(dmel), _p(mel), _p(basis), _p(dspec), B, n_bins, n_mels, frames, _stream()),
          "mel_log_bwd")
    return dspec


def probe_layout(device):
    c = torch.empty(64, 16, dtype=torch.float32, device=device)
    tr = torch.empty(64, 8, dtype=torch.int32, device=device)
    check(_l.get().ttts_probe_mfma_layout(_p(c), _p(tr), _stream()), "probe")
    return c, tr


# ---- FP8 (e4m3) matrix-core GEMMs: the 1 x 1 convolutions of the diffusion step in config #5's arithmetic (csrc/fp8_gemm.hip) ----
def _pad_to(n, m):
    return (n + m - 1) // m * m


_amax_pools = {}        # device key -> [zeroed float32 pool, next free slot]


def fp8_amax(x, out=None):
    """max |x| of the whole tensor as a device scalar (no host read-back).  Without `out` the result lands in a fresh word of a
    pool that was cleared in bulk (one fill launch per 4096 results instead of one per result); every word is handed out once, so
    saved results stay valid for as long as they are referenced.  Under stream capture the word is cleared by the call itself (a
    replay must not start from the previous replay's maximum)."""
    _req(x, torch.float32, "x")
    zero = 0
    if out is None:
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            # (a fill KERNEL, not the library's hipMemsetAsync: the recorded 4-byte memset node left the word uncleared at replay --
            # the diffusion step's fp8 mode replayed from a hipGraph produced NaN losses, eager launches did not)
            out = torch.zeros(1, dtype=torch.float32, device=x.device)
            zero = 1
        else:
            key = (x.device.index, _raw_stream(x.device.index))
            pool = _amax_pools.get(key)
            if pool is None or pool[1] >= pool[0].numel():
                pool = _amax_pools[key] = [torch.zeros(4096, dtype=torch.float32, device=x.device), 0]
            out = pool[0][pool[1]:pool[1] + 1]
            pool[1] += 1
            zero = 1
    check(_l.get().ttts_fp8_amax_f32(_p(x), x.numel(), _p(out), zero, _stream()), "fp8_amax")
    return out


def fp8_quant(x2d, amax, cols_pad=None):
    """x (rows, cols) fp32, reduction axis contiguous -> uint8 (rows, cols_pad) of e4m3 codes, zero padded to a multiple of 64."""
    _req(x2d, torch.float32, "x")
    rows, cols = x2d.shape
    cols_pad = _pad_to(cols, 64) if cols_pad is None else cols_pad
    q = torch.empty(rows, cols_pad, dtype=torch.uint8, device=x2d.device)
    check(_l.get().ttts_fp8_quant_f32(_p(x2d), _p(q), _p(amax), rows, cols, cols_pad, _stream()), "fp8_quant")
    return q


def fp8_quant_transpose(x, amax):
    """x (B, C, T) fp32 -> uint8 (B, T, Cp): the channel (reduction) axis made contiguous, zero padded to a multiple of 64."""
    _req(x, torch.float32, "x")
    B, C, T = x.shape
    Cp = _pad_to(C, 64)
    q = torch.empty(B, T, Cp, dtype=torch.uint8, device=x.device)
    check(_l.get().ttts_fp8_quant_transpose_f32(_p(x), _p(q), _p(amax), B, C, T, Cp, _stream()), "fp8_quant_transpose")
    return q


def fp8_quant_both(x, amax):
    """x (B, C, T) fp32 -> (rows uint8 (B, C, Tp), transposed uint8 (B, T, Cp)) in one pass; Cp, Tp multiples of 64."""
    _req(x, torch.float32, "x")
    B, C, T = x.shape
    Cp, Tp = _pad_to(C, 64), _pad_to(T, 64)
    qr = torch.empty(B, C, Tp, dtype=torch.uint8, device=x.device)
    qt = torch.empty(B, T, Cp, dtype=torch.uint8, device=x.device)
    check(_l.get().ttts_fp8_quant_both_f32(_p(x), _p(qr), _p(qt), _p(amax), B, C, T, Cp, Tp, _stream()), "fp8_quant_both")
    return qr, qt


def fp8_gemm_nt(a, b, y, amax_a, amax_b, M, N, K, bias=None, resid=None, groups_outer=1, groups_inner=1, lda=None, ldb=None,
                a_strides=(0, 0), b_strides=(0, 0), y_strides=(0, None, 1), accumulate=False):
    """Y[go][m][n] (+)= alpha sum_gi sum_k A[go][gi][m][k] B[go][gi][n][k] (+ bias[m]) (+ resid) on the fp8 matrix cores
    (include/ttts_hip.h: ttts_fp8_gemm_nt); a / b uint8 e4m3 codes, y fp32."""
    _req(y, torch.float32, "y")
    ysm = N if y_strides[1] is None else y_strides[1]
    wsb = _l.get().ttts_fp8_gemm_nt_workspace_bytes(M, N, groups_outer, groups_inner)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=y.device) if wsb > 0 else None
    check(_l.get().ttts_fp8_gemm_nt(_p(a), _p(b), _p(y), _p(bias), _p(resid), _p(amax_a), _p(amax_b), M, N, K, groups_outer,
                                    groups_inner, K if lda is None else lda, K if ldb is None else ldb, a_strides[0], a_strides[1],
                                    b_strides[0], b_strides[1], y_strides[0], ysm, y_strides[2], int(accumulate), _p(ws), _stream()),
          "fp8_gemm_nt")
    return y


class Fp8Weight:
    """Both e4m3 layouts of a 1 x 1 convolution weight (Cout, Cin[, 1]) under ONE amax: `q` (Cout, Cinp) for the forward,
    `qt` (Cin, Coutp) for the data gradient."""

    def __init__(self, w):
        Cout, Cin = w.shape[0], w.shape[1]
        self.amax = fp8_amax(w)
        self.q, qt = fp8_quant_both(w.reshape(1, Cout, Cin), self.amax)
        self.q, self.qt = self.q[0], qt[0]


def conv1x1_fp8_fwd(x, w, bias=None, resid=None, wq=None):
    """y[b] = W x[b] (+ bias) (+ resid) with both operands in e4m3 (per-tensor current scaling).  x (B, Cin, T), w (Cout, Cin[, 1]).
    Returns (y, (x rows-layout codes, amax_x), Fp8Weight): what the backward needs -- 1 byte per activation element instead of 4."""
    B, Cin, T = x.shape
    Cout = w.shape[0]
    ax = fp8_amax(x)
    xr, xt = fp8_quant_both(x, ax)                               # (B, Cin, Tp), (B, T, Kp)
    wq = Fp8Weight(w) if wq is None else wq
    Kp = xt.shape[2]
    y = torch.empty(B, Cout, T, dtype=torch.float32, device=x.device)
    fp8_gemm_nt(wq.q, xt, y, wq.amax, ax, Cout, T, Kp, bias=bias, resid=resid, groups_outer=B, b_strides=(T * Kp, 0),
                y_strides=(Cout * T, T, 1))
    return y, (xr, ax), wq


def conv1x1_fp8_bwd(dy, xq, wq, Cin, need_dx=True, dw_out=None):
    """Backward of conv1x1_fp8_fwd from its saved codes: dy (B, Cout, T) is measured and quantised ONCE (both layouts);
    dx[b] = W^T dy[b]; dw_out (Cout, Cin[, 1]) += sum_b dy[b] x[b]^T.  Returns dx (or None)."""
    B, Cout, T = dy.shape
    ady = fp8_amax(dy)
    dyr, dyt = fp8_quant_both(dy, ady)                           # (B, Cout, Tp), (B, T, Coutp)
    dx = None
    if need_dx:
        Kp = dyt.shape[2]
        dx = torch.empty(B, Cin, T, dtype=torch.float32, device=dy.device)
        fp8_gemm_nt(wq.qt, dyt, dx, wq.amax, ady, Cin, T, Kp, groups_outer=B, b_strides=(T * Kp, 0), y_strides=(Cin * T, T, 1))
    if dw_out is not None:
        xr, ax = xq
        Tp = dyr.shape[2]
        fp8_gemm_nt(dyr, xr, dw_out, ady, ax, Cout, Cin, Tp, groups_inner=B, a_strides=(0, Cout * Tp), b_strides=(0, Cin * Tp),
                    y_strides=(0, Cin, 1), accumulate=True)
    return dx
