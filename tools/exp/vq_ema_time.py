"""vq_ema_update at the GPT data path's shape (N = 4096 rows... config #3: 32 x 128 frames = 4096, K = 1024, D = 192): time + parity vs torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from ttts_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, K, D = 4096, 1024, 192
x = torch.randn(N, D, device=dev); idx = torch.randint(0, K, (N,), device=dev)
cs = torch.rand(K, device=dev) * 4; avg = torch.randn(K, D, device=dev); emb = torch.empty(K, D, device=dev)
cs0, avg0 = cs.clone(), avg.clone()
ops.vq_ema_update(x, idx, cs, avg, emb, 0.99, 1e-5)
oh = torch.nn.functional.one_hot(idx, K).float()
cs_ref = cs0 * 0.99 + oh.sum(0) * 0.01
avg_ref = avg0 * 0.99 + (oh.t() @ x) * 0.01
print("cs err %.2e avg err %.2e" % ((cs - cs_ref).abs().max().item(), (avg - avg_ref).abs().max().item()))
def timeit(idx, tag):
    for _ in range(20):
        ops.vq_ema_update(x, idx, cs, avg, emb, 0.99, 1e-5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        ops.vq_ema_update(x, idx, cs, avg, emb, 0.99, 1e-5)
    e1.record(); torch.cuda.synchronize()
    print("vq_ema_update %-28s %.2f us per call (fused + commit), fullest code %d rows" % (tag, e0.elapsed_time(e1) * 1e3 / 200, int(torch.bincount(idx, minlength=K).max())))
timeit(idx, "uniform indices")
cb = torch.randn(K, D, device=dev)
idx2 = ops.vq_nearest(x, cb, want_xq=False)[0]
cs2, avg2 = cs.clone(), avg.clone()
ops.vq_ema_update(x, idx2, cs, avg, emb, 0.99, 1e-5)
oh = torch.nn.functional.one_hot(idx2, K).float()
print("skewed: cs err %.2e avg err %.2e" % ((cs - (cs2 * 0.99 + oh.sum(0) * 0.01)).abs().max().item(), (avg - (avg2 * 0.99 + (oh.t() @ x) * 0.01)).abs().max().item()))
timeit(idx2, "nearest-code indices (bench)")
sys.exit(0)
for _ in range(20):
    ops.vq_ema_update(x, idx, cs, avg, emb, 0.99, 1e-5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(200):
    ops.vq_ema_update(x, idx, cs, avg, emb, 0.99, 1e-5)
e1.record(); torch.cuda.synchronize()
print("vq_ema_update %.2f us per call (fused + commit)" % (e0.elapsed_time(e1) * 1e3 / 200))
