"""Debug helper: forward attention, q128 kernel vs the split kernel (TTTS_FWD_SPLIT) on the same inputs; prints where they differ."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ttts_amd import ops
dev = torch.device("cuda:0")
B, S, H, dh = 1, int(os.environ.get("DBG_S", "200")), 2, 64
p = float(os.environ.get("DBG_P", "0.1"))
D = H * dh
g = torch.Generator(device="cpu").manual_seed(1)
qkv = torch.randn(B, S, 3 * D, generator=g).to(torch.bfloat16).to(dev)
q2 = qkv.view(B * S, 3 * D)
outs = {}
for name, env in (("q128", None), ("split", "1")):
    if env: os.environ["TTTS_FWD_SPLIT"] = env
    else: os.environ.pop("TTTS_FWD_SPLIT", None)
    o = torch.zeros(B, S, D, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B, H, S, device=dev)
    ops.attn_fwd(q2, q2[:, D:], q2[:, 2 * D:], o, lse, B, H, S, dh, (S * 3 * D, 3 * D), (S * D, D), dh ** -0.5, p, 1234567)
    torch.cuda.synchronize()
    outs[name] = (o.float().cpu(), lse.cpu())
a, b = outs["q128"], outs["split"]
d = (a[0] - b[0]).abs().view(S, H, dh).amax(-1)           # [S, H]
print("max |o diff| per head:", d.amax(0).tolist())
bad = (d > 0.05).nonzero()
print("bad (row, head) count", len(bad), "of", S * H)
print("bad rows head0:", [int(r) for r, h in bad.tolist() if h == 0][:80])
print("bad rows head1:", [int(r) for r, h in bad.tolist() if h == 1][:80])
print("lse diff max", (a[1] - b[1]).abs().max().item())
r = int(bad[0][0]) if len(bad) else 0
print("row", r, "q128", a[0].view(S, H, dh)[r, 0, :6].tolist(), "split", b[0].view(S, H, dh)[r, 0, :6].tolist())
