import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ttts_amd import ops
dev = torch.device("cuda:0")
B, S, H, dh = 1, 64, 1, 64
D = H * dh
for scale in (4.0, 6.0, 8.0):
    g = torch.Generator(device="cpu").manual_seed(S)
    qkv = (torch.randn(B, S, 3 * D, generator=g) * scale).to(torch.bfloat16).to(dev)
    o = torch.zeros(B, S, D, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B, H, S, device=dev)
    q2 = qkv.view(B * S, 3 * D)
    ops.attn_fwd(q2, q2[:, D:], q2[:, 2 * D:], o, lse, B, H, S, dh, (S * 3 * D, 3 * D), (S * D, D), dh ** -0.5)
    q, k, v = [t.view(B, S, H, dh).transpose(1, 2) for t in qkv.float().split(D, dim=-1)]
    att = (q @ k.transpose(-1, -2)) * dh ** -0.5
    att = att.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=dev).tril(), float("-inf"))
    rl = torch.logsumexp(att, -1)
    bad = (~torch.isfinite(o.float()).all(-1)).view(-1).nonzero().view(-1).tolist()
    print("scale", scale, "nan rows", bad)
    print("  lse kernel", [round(x, 1) for x in lse.view(-1)[:40].tolist()])
    print("  lse ref   ", [round(x, 1) for x in rl.view(-1)[:40].tolist()])
    am = att.masked_fill(~torch.isfinite(att), -1e9)
    print("  rowmax*log2e by 32-key block:", [(round(float(am[0, 0, r, :32].max()) * 1.4427, 0), round(float(am[0, 0, r, 32:].max()) * 1.4427, 0)) for r in range(32, 44)])
