"""Which part of the diffusion step breaks hipGraph capture?  usage: python diffusion_capture_debug.py <case>"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

from ttts_amd.diffusion.train import DiffusionTrainer
from ttts_amd.diffusion.aa_model import normalize_tacotron_mel

case = sys.argv[1]
dev = torch.device("cuda", 0)
cfg = {"train": {"lr": 1e-4, "timesteps": 1000},
       "aa_diffusion": dict(in_channels=100, out_channels=200, model_channels=512, num_heads=16, num_layers=4, in_latent_channels=512,
                            dropout=0, layer_drop=0.0, unconditioned_percentage=0.0)}
g = torch.Generator().manual_seed(1)
BIG = os.environ.get("DCD_BIG", "0") == "1"
if BIG:
    cfg["aa_diffusion"]["num_layers"] = 6
    mel = (torch.randn(16, 100, 400, generator=g) * 2 - 4).to(dev); ref = (torch.randn(16, 100, 200, generator=g) * 2 - 4).to(dev)
    lat = torch.randn(16, 512, 100, generator=g).to(dev)
else:
    mel = (torch.randn(2, 100, 120, generator=g) * 2 - 4).to(dev); ref = (torch.randn(2, 100, 80, generator=g) * 2 - 4).to(dev)
    lat = torch.randn(2, 512, 30, generator=g).to(dev)
tr = DiffusionTrainer(cfg, device=dev, seed=3)
t = torch.randint(0, 1000, (mel.shape[0],), device=dev); noise = torch.randn_like(mel)
if case.endswith("_sidewarm"):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            tr.train_step(mel, ref, lat)
    torch.cuda.current_stream().wait_stream(side)
else:
    for _ in range(2):
        tr.train_step(mel, ref, lat)
torch.cuda.synchronize()
base = case.replace("_sidewarm", "")
if base.startswith("api"):
    import random
    random.seed(int(base[3:] or 0))
    tr.diffusion.layer_drop = 0.35
    for i in range(int(os.environ.get('DCD_CALLS', '5'))):
        out = tr.train_step_graphed(mel, ref, lat)
        torch.cuda.synchronize()
        print("graphed call", i, float(out["loss"]), sorted(tr._gstate["graphs"]), flush=True)
    print("CASE %s OK" % case, flush=True)
    raise SystemExit(0)
DROP = (1,) if base.endswith("_drop") else ()
base = base.replace("_drop", "")
gr = torch.cuda.CUDAGraph()


def fwd():
    x0 = normalize_tacotron_mel(mel); rf = normalize_tacotron_mel(ref)
    for c in tr._wsplit:
        c.refresh()
    return tr.diffuser.training_losses(tr.diffusion, x0, t, model_kwargs={"latent": lat, "refer": rf, "drop_layers": DROP}, noise=noise)


with torch.cuda.graph(gr):
    if base == "fwd_nograd":
        with torch.no_grad():
            out = fwd()
    elif base == "fwd":
        out = fwd()
    elif base == "fwd_bwd":
        out = fwd()
        tr.optimizer.zero_grad()
        for a in tr._slabs:
            a.begin()
        out["loss_mean"].backward()
        for a in tr._slabs:
            a.reduce()
    elif base == "full":
        out = tr._step_body(mel, ref, lat, t, noise, {"drop_layers": DROP}, False, device_warmup=True)
    else:
        raise SystemExit("unknown case")
for c in tr._wsplit + tr._slabs:
    c.disarm()
gr.replay(); torch.cuda.synchronize()
print("CASE %s OK" % case, flush=True)
