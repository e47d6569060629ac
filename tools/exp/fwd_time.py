import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ttts_amd import ops
dev = torch.device("cuda:0")
B, S, H, dh = 8, 1156, 8, 64
D = H * dh
qkv = torch.randn(B * S, 3 * D, device=dev).to(torch.bfloat16)
o = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B * H * S, device=dev)
def run(p):
    ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, lse, B, H, S, dh, (S * 3 * D, 3 * D), (S * D, D), dh ** -0.5, p, 7)
for p in (0.1, 0.0):
    best = 1e9
    for _ in range(3):
        run(p); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(p)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print("dbg=%s p=%.1f fwd %.1f us" % (os.environ.get("TTTS_ATTN_DBG", "0"), p, best))
