"""Forward / backward attention vs the fp32 torch reference at several shapes and input scales (scores of hundreds of log2 units).
   python tools/attn_scale_check.py      (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttts_amd import ops
dev = torch.device("cuda:0")

def ref(qkv, B, S, H, dh):
    D = H * dh
    q, k, v = [t.view(B, S, H, dh).transpose(1, 2) for t in qkv.float().split(D, dim=-1)]
    att = (q @ k.transpose(-1, -2)) * dh ** -0.5
    causal = torch.ones(S, S, dtype=torch.bool, device=qkv.device).tril()
    att = att.masked_fill(~causal, float("-inf"))
    lse = torch.logsumexp(att, -1)
    pr = torch.softmax(att, -1)
    o = pr.to(torch.bfloat16).float() @ v
    return o.transpose(1, 2).reshape(B, S, D), lse

def rel(a, b): return float((a - b).norm() / b.norm().clamp_min(1e-30))

for (B, S, H, dh, scale) in [(4, 342, 8, 64, 1.0), (4, 342, 8, 64, 4.0), (4, 342, 8, 64, 16.0), (2, 1156, 8, 64, 8.0), (1, 130, 2, 64, 6.0), (3, 64, 4, 64, 6.0)]:
    D = H * dh
    g = torch.Generator(device="cpu").manual_seed(S)
    qkv = (torch.randn(B, S, 3 * D, generator=g) * scale).to(torch.bfloat16).to(dev)
    o = torch.zeros(B, S, D, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B, H, S, device=dev)
    q2 = qkv.view(B * S, 3 * D)
    ops.attn_fwd(q2, q2[:, D:], q2[:, 2 * D:], o, lse, B, H, S, dh, (S * 3 * D, 3 * D), (S * D, D), dh ** -0.5)
    leaf = qkv.float().requires_grad_(True)
    ro, rl = ref(leaf, B, S, H, dh)
    do = torch.randn(B, S, D, generator=g).to(torch.bfloat16).to(dev)
    ro.backward(do.float())
    dqkv = torch.zeros(B * S, 3 * D, dtype=torch.bfloat16, device=dev); ws = torch.empty(B * H * S, device=dev)
    ops.attn_bwd(q2, q2[:, D:], q2[:, 2 * D:], o, do, lse, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], ws, B, H, S, dh, (S * 3 * D, 3 * D), (S * D, D), dh ** -0.5)
    got = dqkv.view(B, S, 3 * D).float()
    print("B%d S%d scale %.0f: o %.2e lse %.2e dq %.2e dk %.2e dv %.2e  finite: o %s dqkv %s" % (
        B, S, scale, rel(o.float(), ro), rel(lse, rl), rel(got[..., :D], leaf.grad[..., :D]), rel(got[..., D:2 * D], leaf.grad[..., D:2 * D]),
        rel(got[..., 2 * D:], leaf.grad[..., 2 * D:]), bool(torch.isfinite(o.float()).all()), bool(torch.isfinite(got).all())))
