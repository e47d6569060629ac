"""Strided forward convolutions of the step's resampling layers: phase-merged (default) against the plain strided launch (flag 4194304)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from ttts_amd import ops
dev = torch.device("cuda", 0)
CASES = [  # B, Cin, Lin, w (Cout, Cin, K), stride, pad
    (32, 16, 163840, (32, 16, 16), 10, 7), (32, 32, 16384, (64, 32, 16), 8, 7), (32, 256, 320, (512, 256, 16), 10, 3),
    (32, 128, 2560, (256, 128, 16), 8, 4), (32, 64, 2048, (96, 64, 8), 2, 3),
]
for B, ci, lin, ws, s, pad in CASES:
    x = torch.randn(B, ci, lin, device=dev); w = torch.randn(ws, device=dev) * 0.05
    res = {}
    for flag in (0, 4194304):
        ops.set_variant_flags(flag)
        y = ops.conv1d_fwd(x, w, None, None, s, pad, 1, in_slope=0.1)
        for _ in range(3):
            ops.conv1d_fwd(x, w, None, None, s, pad, 1, in_slope=0.1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            ops.conv1d_fwd(x, w, None, None, s, pad, 1, in_slope=0.1)
        e1.record(); torch.cuda.synchronize()
        res[flag] = (e0.elapsed_time(e1) * 100, y)
    err = ((res[0][1] - res[4194304][1]).abs().max() / res[4194304][1].abs().max()).item()
    print("fwd x (%d,%d,%d) w %s stride %d: merged %.1f us, plain %.1f us, rel diff %.1e" % (B, ci, lin, ws, s, res[0][0], res[4194304][0], err), flush=True)
ops.set_variant_flags(0)
