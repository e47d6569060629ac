"""Compares the default split-bf16 convolution kernels with the exact f32-MFMA kernels (ttts_conv_ctx.flags = TTTS_CONV_EXACT_F32) on a few path shapes: max relative difference of forward and data gradient.  usage (GPU box): python tools/conv_precision_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ttts_amd import lib, ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def rel(a, b): return ((a - b).abs().max() / b.abs().max()).item()
for (cin, cout, k, s, pad, dil, L, B) in [(64, 64, 11, 1, 25, 5, 2048, 2), (64, 64, 11, 1, 5, 1, 2048, 2), (64, 64, 11, 1, 15, 3, 2048, 2),
                                          (32, 32, 7, 1, 15, 5, 300, 2), (16, 8, 4, 2, 1, 1, 80, 2), (512, 512, 7, 1, 3, 1, 20, 2)]:
    x = torch.randn(B, cin, L, device=dev); w = torch.randn(cout, cin, k, device=dev) * 0.05
    lout = ops.conv_out_len(L, k, s, pad, dil)
    dy = torch.randn(B, cout, lout, device=dev) * 1e-5
    res = torch.randn(B, cin, L, device=dev) * 1e-5
    out = {}
    for flag in (4096, 0):
        ops.set_conv_precision('exact' if flag else 'split_bf16')
        y = ops.conv1d_fwd(x, w, None, None, s, pad, dil, in_slope=0.1)
        dx = ops.conv1d_dgrad(dy, w, L, s, pad, dil, gate=x, gate_slope=0.1, resid=res)
        dx2 = ops.conv1d_dgrad(dy, w, L, s, pad, dil)
        out[flag] = (y.clone(), dx.clone(), dx2.clone())
    ops.set_conv_precision('split_bf16')
    print((cin, cout, k, s, pad, dil, L), "fwd rel %.2e  dgrad(gate,resid) rel %.2e  dgrad plain rel %.2e" % (
        rel(out[0][0], out[4096][0]), rel(out[0][1], out[4096][1]), rel(out[0][2], out[4096][2])))
