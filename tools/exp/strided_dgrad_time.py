"""Strided data gradients of the step's resampling layers: phase-merged launch (default) against the per-phase launches (flag 4194304)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from ttts_amd import ops
dev = torch.device("cuda", 0)
CASES = [  # B, Cout(dy), Lout, w (Cout, Cin, K), Lin, stride, pad
    (32, 32, 16384, (32, 16, 16), 163840, 10, 7), (32, 64, 2048, (64, 32, 16), 16384, 8, 7),
    (32, 512, 32, (512, 256, 16), 320, 10, 3), (32, 256, 320, (256, 128, 16), 2560, 8, 4),
    (32, 96, 1024, (96, 64, 8), 2048, 2, 3),
]
for B, co, lout, ws, lin, s, pad in CASES:
    dy = torch.randn(B, co, lout, device=dev); w = torch.randn(ws, device=dev) * 0.05
    res = {}
    for flag in (0, 4194304):
        ops.set_variant_flags(flag)
        dx = ops.conv1d_dgrad(dy, w, lin, s, pad, 1)
        for _ in range(3):
            ops.conv1d_dgrad(dy, w, lin, s, pad, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            ops.conv1d_dgrad(dy, w, lin, s, pad, 1)
        e1.record(); torch.cuda.synchronize()
        res[flag] = (e0.elapsed_time(e1) * 100, dx)
    err = ((res[0][1] - res[4194304][1]).abs().max() / res[4194304][1].abs().max()).item()
    print("dgrad dy (%d,%d,%d) w %s stride %d: merged %.1f us, per-phase %.1f us, rel diff %.1e" % (B, co, lout, ws, s, res[0][0], res[4194304][0], err), flush=True)
ops.set_variant_flags(0)
