"""Fused cross-attention (csrc/attn_cross.hip) against a torch fp64 reference and against the bgemm + softmax path: forward,
all three gradients, masks (masked_fill -1e4 and -inf), ragged Tq / Tk, d_k 64 / 96 / 128."""
import sys, time, math
import torch
sys.path.insert(0, ".")
from ttts_amd import ops
from ttts_amd.vqvae.attentions import _AttnCoreFn, _AttnFusedFn
dev = torch.device("cuda:0")
torch.manual_seed(0)


def ref(q, k, v, qm, km, H, scale, fill):
    B, C, Tq = q.shape; Tk = k.shape[2]; dk = C // H
    qh = q.double().view(B, H, dk, Tq); kh = k.double().view(B, H, dk, Tk); vh = v.double().view(B, H, dk, Tk)
    s = torch.einsum("bhdt,bhdj->bhtj", qh, kh) * scale
    if qm is not None or km is not None:
        m = (qm if qm is not None else torch.ones(B, Tq, device=dev))[:, None, :, None] * (km if km is not None else torch.ones(B, Tk, device=dev))[:, None, None, :]
        s = s.masked_fill(m == 0, fill)
    p = torch.softmax(s, -1)
    return torch.einsum("bhtj,bhdj->bhdt", p, vh).reshape(B, C, Tq)


for (B, H, dk, Tq, Tk, fill, use_q) in ((32, 4, 128, 256, 100, -1e4, True), (3, 2, 96, 77, 50, -1e4, True), (2, 2, 64, 40, 130, -float("inf"), False),
                                        (4, 4, 128, 256, 256, -1e4, False)):
    C = H * dk
    q = torch.randn(B, C, Tq, device=dev, requires_grad=True); k = torch.randn(B, C, Tk, device=dev, requires_grad=True)
    v = torch.randn(B, C, Tk, device=dev, requires_grad=True)
    km = (torch.arange(Tk, device=dev)[None] < torch.randint(Tk // 2, Tk + 1, (B, 1), device=dev)).float()
    qm = (torch.arange(Tq, device=dev)[None] < torch.randint(Tq // 2, Tq + 1, (B, 1), device=dev)).float() if use_q else None
    scale = 1 / math.sqrt(dk)
    do = torch.randn(B, C, Tq, device=dev)
    out = _AttnFusedFn.apply(q, k, v, qm, km, H, scale, fill)
    out.backward(do)
    g = [t.grad.clone() for t in (q, k, v)]
    for t in (q, k, v): t.grad = None
    r = ref(q, k, v, qm, km, H, scale, fill)
    r.backward(do.double())
    rg = [t.grad.clone() for t in (q, k, v)]
    for t in (q, k, v): t.grad = None
    rel = lambda a, b_: float((a.double() - b_.double()).norm() / b_.double().norm().clamp_min(1e-30))
    print("B%d H%d dk%d Tq%d Tk%d fill %g: out %.2e dq %.2e dk %.2e dv %.2e" % (B, H, dk, Tq, Tk, fill, rel(out, r), rel(g[0], rg[0]), rel(g[1], rg[1]), rel(g[2], rg[2])),
          "finite", all(torch.isfinite(t).all().item() for t in [out] + g))
    if fill > -1e30:
        o2 = _AttnCoreFn.apply(q, k, v, None, None, qm if qm is not None else torch.ones(B, Tq, device=dev), km, H, 0, scale, fill, 0.0, 0)
        o2.backward(do)
        g2 = [t.grad.clone() for t in (q, k, v)]
        for t in (q, k, v): t.grad = None
        print("   vs bgemm path: out %.2e dq %.2e dk %.2e dv %.2e" % (rel(out, o2), rel(g[0], g2[0]), rel(g[1], g2[1]), rel(g[2], g2[2])))
# timing at the MRTE shape
B, H, dk, Tq, Tk = 32, 4, 128, 256, 100
C = H * dk
q = torch.randn(B, C, Tq, device=dev); k = torch.randn(B, C, Tk, device=dev); v = torch.randn(B, C, Tk, device=dev)
km = torch.ones(B, Tk, device=dev); qm = torch.ones(B, Tq, device=dev); do = torch.randn(B, C, Tq, device=dev)
for name, fn in (("fused", lambda: ops.attn_cross_fwd(q, k, v, qm, km, H, 0.088, -1e4)),):
    for _ in range(3): o, lse = fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): o, lse = fn()
    torch.cuda.synchronize(); print(name, "fwd us", (time.time() - t0) / 50 * 1e6)
    for _ in range(3): ops.attn_cross_bwd(q, k, v, qm, km, o, do, lse, H, 0.088, -1e4)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): ops.attn_cross_bwd(q, k, v, qm, km, o, do, lse, H, 0.088, -1e4)
    torch.cuda.synchronize(); print(name, "bwd us", (time.time() - t0) / 50 * 1e6)
qq = q.clone().requires_grad_(True); kk = k.clone().requires_grad_(True); vv = v.clone().requires_grad_(True)
for _ in range(3):
    o2 = _AttnCoreFn.apply(qq, kk, vv, None, None, qm, km, H, 0, 0.088, -1e4, 0.0, 0); o2.backward(do)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(50):
    o2 = _AttnCoreFn.apply(qq, kk, vv, None, None, qm, km, H, 0, 0.088, -1e4, 0.0, 0); o2.backward(do)
torch.cuda.synchronize(); print("bgemm path fwd+bwd us", (time.time() - t0) / 50 * 1e6)
