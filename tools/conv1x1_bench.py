"""1 x 1 convolutions of the VQ-VAE-GAN and diffusion steps: conv1x1_b3_kernel (default) against the operand pre-pass + DMA kernel it
replaces (variant flag 512), forward and data gradient, device time per call from a hipGraph of 20 calls."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ttts_amd import ops

dev = torch.device("cuda", 0)
SHAPES = [  # name, B, Cin, Cout, L
    ("WN res|skip 192->384", 32, 192, 384, 256), ("WN last skip 192->192", 32, 192, 192, 256), ("cond 256->6144", 32, 256, 6144, 1),
    ("enc pre 1025->192", 32, 1025, 192, 256), ("proj 192->384", 32, 192, 384, 256), ("attn q/k/v 192->192", 32, 192, 192, 256),
    ("diff qkv 512->1536 T400", 16, 512, 1536, 400), ("diff proj 512->512 T400", 16, 512, 512, 400), ("diff integ 1024->512 T400", 16, 1024, 512, 400),
    ("diff qkv 512->1536 T100", 16, 512, 1536, 100), ("diff proj 512->512 T232", 16, 512, 512, 232),
]


def graph_time_us(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


for name, B, cin, cout, L in SHAPES:
    x = torch.randn(B, cin, L, device=dev); w = torch.randn(cout, cin, 1, device=dev) * 0.05
    dy = torch.randn(B, cout, L, device=dev)
    out = {}
    for flag in (0, 512):
        ops.set_variant_flags(flag)
        out[flag] = (graph_time_us(lambda: ops.conv1d_fwd(x, w, None, None, 1, 0, 1, in_slope=0.1)),
                     graph_time_us(lambda: ops.conv1d_dgrad(dy, w, L)))
    ops.set_variant_flags(0)
    gf = 2.0 * B * cin * cout * L / 1e9
    print("%-28s %6.2f GFLOP   fwd %6.1f us (general path %6.1f)   dgrad %6.1f us (general %6.1f)   fwd %5.1f TF/s"
          % (name, gf, out[0][0], out[512][0], out[0][1], out[512][1], gf / out[0][0] * 1e-3 * 1e3 / 1e3 * 1e3), flush=True)
