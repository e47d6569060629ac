"""The body of tests/test_gpu_diffusion.py::test_graphed_diffusion_step_follows_the_eager_step, stand-alone, with switches."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

from ttts_amd import ops
from ttts_amd.diffusion.train import DiffusionTrainer

if os.environ.get("D2_EXACT", "0") == "1":
    ops.set_conv_precision("exact")
dev = torch.device("cuda", 0)
cfg = {"train": {"lr": 1e-4, "timesteps": 1000},
       "aa_diffusion": dict(in_channels=100, out_channels=200, model_channels=512, num_heads=16, num_layers=4, in_latent_channels=512,
                            dropout=0, layer_drop=float(os.environ.get('D2_LD', '0.35')), unconditioned_percentage=float(os.environ.get("D2_UNCOND", "0.0")))}
g = torch.Generator().manual_seed(1)
mel = (torch.randn(2, 100, 120, generator=g) * 2 - 4).to(dev); ref = (torch.randn(2, 100, 80, generator=g) * 2 - 4).to(dev)
lat = torch.randn(2, 512, 30, generator=g).to(dev)
keep = []
if os.environ.get('D2_USEGEN', '0') == '1':
    _ = torch.rand(4, device=dev)
for graphed in ((False, True) if os.environ.get("D2_TWO", "1") == "1" else (True,)):
    if os.environ.get('D2_NOSEED', '0') != '1':
        random.seed(11); torch.manual_seed(11)
    tr = DiffusionTrainer(cfg, device=dev, seed=3)
    keep.append(tr)
    if os.environ.get("D2_PROJ", "1") == "1":
        with torch.no_grad():
            for k, p in tr.diffusion.named_parameters():
                if k.endswith("proj_out.weight"):
                    p.normal_(0, 0.02)
    if os.environ.get('D2_LD_LATER', '0') == '1':
        tr.diffusion.layer_drop = 0.35
    for i in range(8):
        fn = tr.train_step_graphed if (graphed and i >= 2) else tr.train_step
        out = fn(mel, ref, lat)
        torch.cuda.synchronize()
        print(graphed, i, float(out["loss"]), flush=True)
print("CASE OK", flush=True)
