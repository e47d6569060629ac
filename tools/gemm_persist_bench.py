#!/usr/bin/env python3
"""Persistent wave-specialised NT GEMM vs the one-tile-per-workgroup kernel at the GPT step's shapes (HIP-event timing).
   python tools/gemm_persist_bench.py      (GPU box)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ttts_amd import ops  # noqa: E402
from ttts_amd.lib import EPI_DGELU_BF16, EPI_GELU_BF16, EPI_RESID_ADD_F32, EPI_STORE_BF16  # noqa: E402

dev = torch.device("cuda:0")
REPS, ROUNDS = int(os.environ.get("KB_REPS", "20")), int(os.environ.get("KB_ROUNDS", "3"))


def timeit(fn):
    best = []
    for _ in range(ROUNDS):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / REPS * 1e3)
    return min(best)


out = {}
M = 9248
ws = ops.gemm_nt_workspace(dev)
for name, N, K, epi in (("c_attn fwd", 1536, 512, EPI_STORE_BF16), ("c_fc gelu", 2048, 512, EPI_GELU_BF16),
                        ("mlp c_proj resid", 512, 2048, EPI_RESID_ADD_F32), ("attn c_proj resid", 512, 512, EPI_RESID_ADD_F32),
                        ("dgelu", 2048, 512, EPI_DGELU_BF16), ("dX c_fc", 512, 2048, EPI_STORE_BF16), ("dX c_attn", 512, 1536, EPI_STORE_BF16),
                        ("mel_head", 1026, 512, EPI_STORE_BF16)):
    Mr = 8208 if name == "mel_head" else M
    a = torch.randn(Mr, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    ldc = (N + 7) // 8 * 8
    cdt = torch.float32 if epi == EPI_RESID_ADD_F32 else torch.bfloat16
    c = torch.zeros(Mr, ldc, dtype=cdt, device=dev)
    aux = torch.zeros(Mr, ldc, dtype=torch.bfloat16, device=dev) if epi in (EPI_GELU_BF16, EPI_DGELU_BF16) else None
    rin = torch.randn(Mr, ldc, device=dev) if epi == EPI_RESID_ADD_F32 else None
    fl = 2.0 * Mr * N * K
    row = {}
    for tag, w in (("one_tile", None), ("persistent", ws)):
        us = timeit(lambda: ops.gemm_nt(a, b, c, bias, aux=aux, epilogue=epi, resid_in=rin, n=N, workspace=w))
        row[tag] = "%.1f us  %.0f TF/s" % (us, fl / us / 1e6)
    out["%s M%d N%d K%d" % (name, Mr, N, K)] = row
print(json.dumps(out, indent=1))
