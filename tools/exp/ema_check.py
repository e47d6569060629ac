import sys, torch, numpy as np
sys.path.insert(0, ".")
from ttts_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for N, K, D in ((4096, 1024, 192), (8192, 1024, 192), (300, 40, 64), (4096, 1024, 192)):
    x = torch.randn(N, D, device=dev); idx = torch.randint(0, K, (N,), device=dev)
    if N == 4096 and K == 1024: idx[:2000] = 7      # a crowded code
    cs = torch.rand(K, device=dev) * 5; avg = torch.randn(K, D, device=dev); emb = torch.zeros(K, D, device=dev)
    cs0, avg0 = cs.clone(), avg.clone()
    ops.vq_ema_update(x, idx, cs, avg, emb, 0.99, 1e-5)
    cnt = torch.bincount(idx, minlength=K).float()
    sums = torch.zeros(K, D, device=dev, dtype=torch.float64).index_add_(0, idx, x.double())
    cs_ref = cs0 * 0.99 + cnt * 0.01
    avg_ref = (avg0.double() * 0.99 + sums * 0.01).float()
    tot = cs_ref.double().sum().float()
    sm = (cs_ref + 1e-5) / (tot + K * 1e-5) * tot
    emb_ref = avg_ref / sm[:, None]
    print(N, K, D, "cs", float((cs - cs_ref).abs().max()), "avg", float((avg - avg_ref).abs().max()), "emb rel", float(((emb - emb_ref).abs() / (emb_ref.abs() + 1e-3)).max()))
    # determinism
    cs2, avg2, emb2 = cs0.clone(), avg0.clone(), torch.zeros_like(emb)
    ops.vq_ema_update(x, idx, cs2, avg2, emb2, 0.99, 1e-5)
    print("  deterministic:", torch.equal(cs, cs2) and torch.equal(avg, avg2) and torch.equal(emb, emb2))
import time
x = torch.randn(4096, 192, device=dev); idx = torch.randint(0, 1024, (4096,), device=dev)
cs = torch.rand(1024, device=dev) * 5; avg = torch.randn(1024, 192, device=dev); emb = torch.zeros(1024, 192, device=dev)
for _ in range(5): ops.vq_ema_update(x, idx, cs, avg, emb)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(200): ops.vq_ema_update(x, idx, cs, avg, emb)
torch.cuda.synchronize(); print("us per call (eager, incl. launch overheads)", (time.time() - t0) / 200 * 1e6)
