"""Capture ONE forced layer-drop pattern after two eager steps (eager pattern from D3_EAGER, e.g. "" or "2"): which patterns crash?"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

from ttts_amd.diffusion.train import DiffusionTrainer

pat = tuple(int(c) for c in sys.argv[1].split(",") if c)
eager = tuple(int(c) for c in os.environ.get("D3_EAGER", "").split(",") if c)
dev = torch.device("cuda", 0)
cfg = {"train": {"lr": 1e-4, "timesteps": 1000},
       "aa_diffusion": dict(in_channels=100, out_channels=200, model_channels=512, num_heads=16, num_layers=4, in_latent_channels=512,
                            dropout=0, layer_drop=0.0, unconditioned_percentage=0.0)}
g = torch.Generator().manual_seed(1)
mel = (torch.randn(2, 100, 120, generator=g) * 2 - 4).to(dev); ref = (torch.randn(2, 100, 80, generator=g) * 2 - 4).to(dev)
lat = torch.randn(2, 512, 30, generator=g).to(dev)
tr = DiffusionTrainer(cfg, device=dev, seed=3)
for _ in range(2):
    tr.train_step(mel, ref, lat, inject={"drop_layers": eager})
torch.cuda.synchronize()
t = torch.randint(0, 1000, (2,), device=dev); noise = torch.randn_like(mel)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = tr._step_body(mel, ref, lat, t, noise, {"drop_layers": pat}, False, device_warmup=True)
gr.replay(); torch.cuda.synchronize()
print("PATTERN %s after eager %s OK %.4f" % (pat, eager, float(out["loss"])), flush=True)
