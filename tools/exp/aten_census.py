"""Which Python lines / autograd nodes of the VQ-VAE-GAN step still issue ATen kernels (add / copy / fill / mul ...)?
One profiled step (CPU-side events with stacks); prints a census keyed by (op, shapes, nearest ttts_amd frame or parent node)."""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from ttts_amd.vqvae.train import SyntheticVqvaeBatches, VqvaeTrainer, get_hparams
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "vqvae"
if which == "vqvae":
    tr = VqvaeTrainer(get_hparams())
    cb = tr.net_g.quantizer.vq.layers[0]._codebook
    with torch.no_grad():
        cb.inited.fill_(1); cb.embed.normal_(0, 0.3); cb.embed_avg.copy_(cb.embed * 4); cb.cluster_size.fill_(4.0)
    data = next(iter(SyntheticVqvaeBatches(B, device=tr.device)))
    step = lambda: tr.train_step(data)
else:
    from ttts_amd.diffusion.train import DiffusionTrainer
    cfg = {"train": {"lr": 1e-4, "timesteps": 1000},
           "aa_diffusion": dict(in_channels=100, out_channels=200, model_channels=512, num_heads=16, num_layers=6, in_latent_channels=512,
                                dropout=0, layer_drop=0.1)}
    tr = DiffusionTrainer(cfg, device="cuda:0")
    g = torch.Generator().manual_seed(0)
    mel = (torch.randn(16, 100, 400, generator=g) * 2 - 4).cuda(); ref = (torch.randn(16, 100, 200, generator=g) * 2 - 4).cuda()
    lat = torch.randn(16, 512, 100, generator=g).cuda()
    step = lambda: tr.train_step(mel, ref, lat)
from ttts_amd.vqvae import attentions as _att
_next = _att._SeedSource.next.__func__
_att._SeedSource.next = classmethod(lambda cls: _next(cls) & 0x7FFFFFFFFFFFFFFF)   # (record_shapes cannot box a Python int >= 2^63)
for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = prof.events()
census = collections.Counter()
ktime = collections.Counter()
def frame_of(e):
    for fr in (e.stack or []):
        if "ttts_amd" in fr and "ops.py" not in fr:
            return fr.split("ttts_amd/")[-1]
    for fr in (e.stack or []):
        if "ttts_amd" in fr:
            return fr.split("ttts_amd/")[-1]
    p = e.cpu_parent
    while p is not None:
        if "evaluate_function" in p.name or "Backward" in p.name:
            return p.name[:80]
        p = p.cpu_parent
    return "?"
for e in evs:
    if not e.name.startswith("aten::"):
        continue
    # only leaf ATen ops that launch a kernel
    kt = sum(k.duration for k in e.kernels) if getattr(e, "kernels", None) else 0
    if not e.kernels:
        continue
    shapes = str([s for s in (e.input_shapes or []) if s])[:70]
    key = (e.name, shapes, frame_of(e))
    census[key] += 1
    ktime[key] += kt
rows = sorted(census.items(), key=lambda kv: -ktime[kv[0]])
tot = sum(ktime.values())
print("ATEN kernels: %d launches, %.2f ms device time" % (sum(census.values()), tot / 1e3))
for (name, shapes, fr), n in rows[:120]:
    print("%5d x %-22s %8.1f us  %-70s %s" % (n, name, ktime[(name, shapes, fr)], shapes, fr))
