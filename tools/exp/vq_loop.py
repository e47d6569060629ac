import sys, torch
sys.path.insert(0, "/root/repo")
from ttts_amd import ops
x = torch.randn(4096, 192, device="cuda"); cb = torch.randn(1024, 192, device="cuda")
for _ in range(30): ops.vq_nearest(x, cb)
torch.cuda.synchronize()
