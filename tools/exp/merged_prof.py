"""Two phase-merged data gradients alone (for a kernel-trace profile): the encoders' s10 layer and DiscriminatorP's 512 -> 1024 s3 layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from ttts_amd import ops
dev = torch.device("cuda", 0)
for B, co, lout, ws, lin, s, pad in [(32, 32, 16384, (32, 16, 16), 163840, 10, 7), (448, 1024, 37, (1024, 512, 5), 109, 3, 2)]:
    dy = torch.randn(B, co, lout, device=dev); w = torch.randn(ws, device=dev) * 0.05
    for flag in (0, 4194304):
        ops.set_variant_flags(flag)
        for _ in range(5):
            ops.conv1d_dgrad(dy, w, lin, s, pad, 1)
        torch.cuda.synchronize()
ops.set_variant_flags(0)
