"""Micro-benchmark of the conv1d family at shapes of the VQ-VAE-GAN step, B = 8.
usage: python tools/conv_bench.py [flags]   (flags -> ttts_conv_ctx.flags, e.g. 256 = direct kernels only, 4096 = exact fp32)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as ge

ge.build()
from ttts_amd import lib, ops

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ops.set_variant_flags(flags)
dev = torch.device("cuda", 0)
B = int(os.environ.get("CB_B", "8"))
SHAPES = [  # name, Cin, Cout, K, stride, pad, dil, L
    ("dec RB1(128) k11 d1", 128, 128, 11, 1, 5, 1, 2560),
    ("dec RB1(128) k11 d5", 128, 128, 11, 1, 25, 5, 2560),
    ("dec RB1(256) k7 d3", 256, 256, 7, 1, 9, 3, 320),
    ("dec RB1(64) k7 d1", 64, 64, 7, 1, 3, 1, 5120),
    ("dec RB1(32) k11 d1", 32, 32, 11, 1, 5, 1, 10240),
    ("dec RB1(16) k7 d1", 16, 16, 7, 1, 3, 1, 20480),
    ("enc RB1(32) k7 d3", 32, 32, 7, 1, 9, 3, 16384),
    ("WN in 192->384 k5", 192, 384, 5, 1, 2, 1, 256),
    ("WN rs 192->384 k1", 192, 384, 1, 1, 0, 1, 256),
    ("FFN 192->768 k3", 192, 768, 3, 1, 1, 1, 256),
    ("DiscP 1024->1024 k5 (p=3)", 1024, 1024, 5, 1, 2, 1, 85),
    ("DiscP 512->1024 k5 s3 (p=3)", 512, 1024, 5, 3, 2, 1, 253),
    ("DiscS 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 80),
    ("pre 1025->192 k1", 1025, 192, 1, 1, 0, 1, 256),
]


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


rows = []
ONLY = os.environ.get("CB_ONLY", "")          # substring filter on the shape name
for name, cin, cout, k, s, pad, dil, L in SHAPES:
    if ONLY and ONLY not in name:
        continue
    bb = B * 3 if "p=3" in name else B
    x = torch.randn(bb, cin, L, device=dev); w = torch.randn(cout, cin, k, device=dev) * 0.05
    lout = ops.conv_out_len(L, k, s, pad, dil)
    dy = torch.randn(bb, cout, lout, device=dev)
    fl = 2.0 * bb * cin * cout * k * lout
    t_f = timeit(lambda: ops.conv1d_fwd(x, w, None, None, s, pad, dil, in_slope=0.1))
    t_d = timeit(lambda: ops.conv1d_dgrad(dy, w, L, s, pad, dil, gate=x, gate_slope=0.1))
    t_w = timeit(lambda: ops.conv1d_wgrad(dy, x, k, s, pad, dil, x_slope=0.1))
    rows.append({"shape": name, "gflop": fl / 1e9, "fwd_us": t_f * 1e6, "fwd_tf": fl / t_f / 1e12, "dgrad_us": t_d * 1e6,
                 "dgrad_tf": fl / t_d / 1e12, "wgrad_us": t_w * 1e6, "wgrad_tf": fl / t_w / 1e12})
    print("%-30s %7.2f GF | fwd %8.1f us %6.1f TF/s | dgrad %8.1f us %6.1f TF/s | wgrad %8.1f us %6.1f TF/s" % (
        name, fl / 1e9, t_f * 1e6, fl / t_f / 1e12, t_d * 1e6, fl / t_d / 1e12, t_w * 1e6, fl / t_w / 1e12), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"flags": flags, "batch": B, "rows": rows}, open("gpurun_out/conv_bench_%d.json" % flags, "w"), indent=1)
