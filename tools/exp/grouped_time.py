"""DiscriminatorS grouped layers (cig = 4): forward / weight-gradient times at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from ttts_amd import ops
dev = torch.device("cuda", 0)
def t(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100
for B, cin, lin, cout, G in [(32, 1024, 320, 1024, 256), (64, 1024, 320, 1024, 256), (32, 256, 1280, 1024, 64), (64, 64, 5120, 256, 16), (64, 16, 20480, 64, 4)]:
    x = torch.randn(B, cin, lin, device=dev); w = torch.randn(cout, cin // G, 41, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    y = ops.conv1d_fwd(x, w, b, None, 4, 20, 1, out_act="lrelu", out_slope=0.1, groups=G)
    dy = torch.randn_like(y); dw = torch.zeros_like(w)
    print("g%d (%d,%d,%d): fwd %.1f us, wgrad %.1f us, dgrad %.1f us" % (G, B, cin, lin,
          t(lambda: ops.conv1d_fwd(x, w, b, None, 4, 20, 1, out_act="lrelu", out_slope=0.1, groups=G)),
          t(lambda: ops.conv1d_wgrad(dy, x, 41, 4, 20, 1, out=dw, groups=G)),
          t(lambda: ops.conv1d_dgrad(dy, w, lin, 4, 20, 1, groups=G))), flush=True)
