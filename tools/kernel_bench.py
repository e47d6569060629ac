#!/usr/bin/env python3
"""Per-kernel micro-benchmarks with HIP-event timing (interleaved rounds), incl. timing ablations of gemm_nt.
   python tools/kernel_bench.py gemm | attn | all      (GPU box)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ttts_amd import lib, ops  # noqa: E402
from ttts_amd.lib import EPI_DGELU_BF16, EPI_GELU_BF16, EPI_RESID_ADD_F32, EPI_STORE_BF16  # noqa: E402

dev = torch.device("cuda:0")


REPS = int(os.environ.get('KB_REPS', '20'))
ROUNDS = int(os.environ.get('KB_ROUNDS', '3'))


def timeit(fn, reps=None, rounds=None):
    reps = reps or REPS
    rounds = rounds or ROUNDS
    best = []
    for _ in range(rounds):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
    return min(best)


def bench_gemm():
    out = {}
    M = 9248
    for name, N, K, epi in (("c_attn fwd", 1536, 512, EPI_STORE_BF16), ("c_fc gelu", 2048, 512, EPI_GELU_BF16),
                            ("mlp c_proj resid", 512, 2048, EPI_RESID_ADD_F32), ("attn c_proj resid", 512, 512, EPI_RESID_ADD_F32),
                            ("dgelu", 2048, 512, EPI_DGELU_BF16), ("dX c_fc", 512, 2048, EPI_STORE_BF16)):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        cdt = torch.float32 if epi == EPI_RESID_ADD_F32 else torch.bfloat16
        c = torch.zeros(M, N, dtype=cdt, device=dev)
        aux = torch.zeros(M, N, dtype=torch.bfloat16, device=dev) if epi in (EPI_GELU_BF16, EPI_DGELU_BF16) else None
        rin = torch.randn(M, N, device=dev) if epi == EPI_RESID_ADD_F32 else None
        fl = 2.0 * M * N * K
        us = timeit(lambda: ops.gemm_nt(a, b, c, bias, aux=aux, epilogue=epi, resid_in=rin))
        out[name + " M%d N%d K%d" % (M, N, K)] = "%.1f us  %.0f TF/s" % (us, fl / us / 1e6)
    Kr = 9280      # the engine zero-pads the reduction rows to a multiple of 64
    for name, Mo, No in (("dW c_attn", 512, 1536), ("dW c_fc", 512, 2048), ("dW mlp c_proj", 2048, 512), ("dW attn c_proj", 512, 512),
                         ("dW mel_head", 1032, 512)):
        at = torch.randn(Kr, (Mo + 7) // 8 * 8, device=dev).to(torch.bfloat16)
        bt = torch.randn(Kr, No, device=dev).to(torch.bfloat16)
        c = torch.zeros(Mo, No, device=dev)
        ws = ops.gemm_tn_workspace(Mo, No, Kr, dev)
        us = timeit(lambda: ops.gemm_tn_accum(at, bt, c, workspace=ws))
        out[name] = "%.1f us  %.0f TF/s (incl. slab reduce)" % (us, 2.0 * Kr * Mo * No / us / 1e6)
    return out


def bench_mem():
    out = {}
    for mb in (38, 75, 300):
        n = mb * 1024 * 1024 // 4
        a = torch.empty(n, device=dev)
        b = torch.empty(n, device=dev)
        us = timeit(lambda: a.fill_(1.0))
        out["fill %d MB" % mb] = "%.1f us  %.2f TB/s" % (us, n * 4 / us / 1e6)
        us = timeit(lambda: b.copy_(a))
        out["copy %d MB" % mb] = "%.1f us  %.2f TB/s (r+w)" % (us, 2 * n * 4 / us / 1e6)
    return out


def bench_attn():
    out = {}
    B, S, H, dh = 8, 1156, 8, 64
    D = H * dh
    qkv = torch.randn(B * S, 3 * D, device=dev).to(torch.bfloat16)
    o = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev)
    do = torch.randn(B * S, D, device=dev).to(torch.bfloat16)
    lse = torch.zeros(B * H * S, device=dev)
    dqkv = torch.zeros_like(qkv)
    ws = torch.zeros(B * H * S, device=dev)
    fl = 4.0 * B * H * dh * S * (S + 1) / 2
    for p in (0.0, 0.1):
        us = timeit(lambda: ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, lse, B, H, S, dh, (S * 3 * D, 3 * D), (S * D, D), dh ** -0.5, p, 7))
        out["fwd p=%.1f" % p] = "%.1f us  %.0f TF/s (causal-useful)" % (us, fl / us / 1e6)
        us = timeit(lambda: ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, do, lse, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], ws, B, H, S, dh,
                                         (S * 3 * D, 3 * D), (S * D, D), dh ** -0.5, p, 7))
        out["bwd p=%.1f" % p] = "%.1f us  %.0f TF/s (causal-useful, 5 matmuls)" % (us, 2.5 * fl / us / 1e6)
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    res = {}
    if which in ("gemm", "all"):
        res["gemm"] = bench_gemm()
    if which in ("mem", "gemm", "all"):
        res["mem"] = bench_mem()
    if which in ("attn", "all"):
        res["attn"] = bench_attn()
    print(json.dumps(res, indent=1))
