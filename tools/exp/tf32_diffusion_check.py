"""Controlled comparison of the diffusion step's gradients in the default (split-bf16) and the 'tf32class' convolution modes: same
weights, inputs, timesteps, noise, no dropped layers; per-tensor relative L2 and the global norms (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ttts_amd import ops
from ttts_amd.diffusion.train import DiffusionTrainer

dev = torch.device("cuda", 0)
acfg = dict(in_channels=100, out_channels=200, model_channels=512, num_heads=16, num_layers=6, in_latent_channels=512, dropout=0, layer_drop=0.1)
B = 4
g = torch.Generator().manual_seed(6)
x0 = torch.tanh(torch.randn(B, 100, 400, generator=g) * 0.7).to(dev); refer = torch.tanh(torch.randn(B, 100, 200, generator=g) * 0.7).to(dev)
latent = torch.randn(B, 512, 100, generator=g).to(dev); t = torch.tensor([0, 641, 100, 900]).to(dev); noise = torch.randn(B, 100, 400, generator=g).to(dev)
res = {}
for mode in ("split_bf16", "tf32class"):
    torch.manual_seed(0)
    tr = DiffusionTrainer({"train": {"lr": 1e-4, "timesteps": 1000}, "aa_diffusion": acfg}, device=dev)
    with torch.no_grad():
        for k, p in tr.diffusion.named_parameters():
            if k.endswith("proj_out.weight"):
                p.normal_(0, 0.02)
    ops.set_conv_precision(mode)
    box = {}
    orig = tr.optimizer.step
    def spy(*a, _tr=tr, _o=orig, **kw):
        box["g"] = _tr.optimizer.flat_g.clone(); return _o(*a, **kw)
    tr.optimizer.step = spy
    out = tr.train_step(x0, refer, latent, t=t, noise=noise, inject={"uncond": torch.zeros(B, dtype=torch.bool, device=dev), "drop_layers": set()}, normalized=True)
    torch.cuda.synchronize()
    res[mode] = (box["g"].double().cpu(), float(out["loss"]), tr)
    ops.set_conv_precision("split_bf16")
ga, la, tr = res["split_bf16"]; gb, lb, _ = res["tf32class"]
print("loss", la, lb, "global norm", float(ga.norm()), float(gb.norm()), "rel L2 of the whole arena", float((ga - gb).norm() / ga.norm()))
names = {p.data_ptr(): k for k, p in tr.diffusion.named_parameters()}
worst = []
for p, o in zip(tr.optimizer.params, tr.optimizer.offsets):
    a, b = ga[o:o + p.numel()], gb[o:o + p.numel()]
    if float(a.norm()) > 1e-9:
        worst.append((float((a - b).norm() / a.norm()), float(a.norm()), names[p.data_ptr()]))
worst.sort(reverse=True)
for w in worst[:12]:
    print("  rel %.3e  norm %.3e  %s" % w)
