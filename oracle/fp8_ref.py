"""ORACLE (test infrastructure only -- imported by tests/, never by the product path).

CPU restatement of the fp8 (OCP e4m3) arithmetic of csrc/fp8_gemm.hip: per-tensor current scaling (scale = 448 / amax of the very
tensor, amax == 0 -> 1), round-to-nearest-even conversion to e4m3 after clamping to +-448 (torch.float8_e4m3fn's cast is the same
IEEE-style rounding the hardware's v_cvt_pk_fp8_f32 performs on in-range values), exact fp32 products of the e4m3 values, fp32
accumulation, result scaled by amax_a amax_b / 448^2.  What it stands for in the reference: the nn.Conv1d(k = 1) calls of
ttts/utils/utils.py:172-215 (AttentionBlock.qkv / .proj_out), ttts/diffusion/aa_model.py:70-131 (ResBlock.in_layers[2]) and :228
(integrating_conv) under BASELINE config #5's "fp8 MFMA GEMMs" -- the reference itself runs them in fp32 / bf16 autocast, so this
oracle pins the KERNEL's arithmetic (bit-level, up to summation order), while tests/test_gpu_diffusion.py holds the fp8 step to a
stated tolerance against the reference-generated fixture tests/golden/diffusion.npz.

Parity status: the quantiser is pinned against torch.float8_e4m3fn (an independent implementation of the OCP format); there is no
reference-side fp8 code to pin against (the reference has none)."""
import torch

FP8_MAX = 448.0


def scale_of(x):
    a = x.abs().max().float()
    return (FP8_MAX / a) if float(a) > 0 else torch.tensor(1.0), a


def quant(x):
    """-> (e4m3 values as fp32, amax)."""
    s, a = scale_of(x)
    q = (x.float() * s).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float()
    return q, a


def alpha(amax_a, amax_b):
    fa = (amax_a if float(amax_a) > 0 else torch.tensor(FP8_MAX)) / FP8_MAX
    fb = (amax_b if float(amax_b) > 0 else torch.tensor(FP8_MAX)) / FP8_MAX
    return float(fa) * float(fb)


def conv1x1_fwd(x, w, bias=None, resid=None):
    """x (B, Cin, T), w (Cout, Cin): y[b] = W x[b] (+ bias) (+ resid) with both operands quantised per tensor."""
    xq, ax = quant(x)
    wq, aw = quant(w.reshape(w.shape[0], -1))
    y = torch.einsum("oc,bct->bot", wq.double(), xq.double()).float() * alpha(aw, ax)
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    if resid is not None:
        y = y + resid
    return y


def conv1x1_dgrad(dy, w):
    dq, ad = quant(dy)
    wq, aw = quant(w.reshape(w.shape[0], -1))
    return torch.einsum("oc,bot->bct", wq.double(), dq.double()).float() * alpha(aw, ad)


def conv1x1_wgrad(dy, x):
    dq, ad = quant(dy)
    xq, ax = quant(x)
    return torch.einsum("bot,bct->oc", dq.double(), xq.double()).float() * alpha(ad, ax)
