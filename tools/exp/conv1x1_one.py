"""One 1 x 1 convolution shape, forward, N launches (for rocprofv3 --pmc passes): python tools/exp/conv1x1_one.py B Cin Cout T [n]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge; ge.build()
from ttts_amd import ops
B, cin, cout, T = [int(v) for v in sys.argv[1:5]]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda", 0)
x = torch.randn(B, cin, T, device=dev); w = torch.randn(cout, cin, 1, device=dev) * 0.05
for _ in range(n):
    y = ops.conv1d_fwd(x, w)
torch.cuda.synchronize()
print("done", float(y.abs().mean()))
