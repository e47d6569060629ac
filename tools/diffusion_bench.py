"""Diffusion mel-denoiser train step at BASELINE config #5: x_start (B,100,400), latent (B,512,100), refer (B,100,200),
AA_diffusion per ttts/diffusion/config.yaml, B = 16 (DFB_B), fp32 with split-bf16 matrix-core convolutions.  Reports steps/s,
mel frames/s and an algorithmic-FLOP rate (3 x forward FLOPs of convs + attention)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ttts_amd.diffusion.train import DiffusionTrainer  # noqa: E402

B, STEPS, WARM = int(os.environ.get("DFB_B", 16)), int(os.environ.get("DFB_STEPS", 5)), int(os.environ.get("DFB_WARMUP", 2))
cfg = {"train": {"lr": 1e-4, "timesteps": 1000},
       "aa_diffusion": dict(in_channels=100, out_channels=200, model_channels=512, num_heads=16, num_layers=6, in_latent_channels=512,
                            dropout=0, layer_drop=0.1)}
tr = DiffusionTrainer(cfg, device="cuda:0")
with torch.no_grad():      # the reference zero-initialises every attention output projection: give them signal
    for k, p in tr.diffusion.named_parameters():
        if k.endswith("proj_out.weight"):
            p.normal_(0, 0.02)
g = torch.Generator().manual_seed(0)
mel = (torch.randn(B, 100, 400, generator=g) * 2 - 4).cuda(); ref = (torch.randn(B, 100, 200, generator=g) * 2 - 4).cuda()
lat = torch.randn(B, 512, 100, generator=g).cuda()
for _ in range(WARM):
    out = tr.train_step(mel, ref, lat)
torch.cuda.synchronize(); t0 = time.perf_counter()
step_fn = tr.train_step_graphed if os.environ.get("DFB_GRAPH", "0") == "1" else tr.train_step
if step_fn is not tr.train_step:
    for _ in range(5):                     # (the first call records the frequent layer-drop patterns)
        out = step_fn(mel, ref, lat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(STEPS):
    out = step_fn(mel, ref, lat)
host_issue = (time.perf_counter() - t0) / STEPS          # the host's share: launches queued, device not yet waited for
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / STEPS
C, T, Tl, Tr = 512, 400, 100, 200


def attn(t):      # qkv + proj 1x1 convs, QK^T and PV
    return 2 * t * C * 3 * C + 2 * t * C * C + 4 * t * t * C


def resb(t):
    return 2 * t * C * C + 2 * t * C * C * 3


fwd = (2 * Tl * 512 * C * 3 + 3 * attn(Tl)) + (2 * Tr * 100 * C * 3 + 3 * attn(Tr) + 2 * (Tr + 32) * C * C * 3 + 4 * attn(Tr + 32)) \
    + 3 * (resb(T) + attn(T)) + 2 * T * 100 * C * 3 + 2 * T * 2 * C * C + 6 * (resb(T) + attn(T)) + 3 * resb(T) + 2 * T * C * 200 * 3
print(json.dumps({"B": B, "ms_per_step": round(dt * 1e3, 2), "host_issue_ms": round(host_issue * 1e3, 2), "steps_per_s": round(1 / dt, 3), "mel_frames_per_s": round(B * 400 / dt, 1),
                  "fwd_GFLOP_per_sample": round(fwd / 1e9, 2), "algorithmic_TFLOPs": round(3 * fwd * B / dt / 1e12, 2),
                  "loss": float(out["loss"]), "grad_norm": float(out["grad_norm"])}))
