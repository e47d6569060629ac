"""PosteriorAudioEncoder backward under run_branches: which gradients differ from the sequential run?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from ttts_amd import ops
from ttts_amd.vqvae.vq2 import PosteriorAudioEncoder
from oracle import vqvae_ref
dev = torch.device("cuda", 0)
ops.set_conv_precision("exact")
g = np.load("tests/golden/vqvae_flow.npz")
D = lambda k: torch.from_numpy(g[k]).to(dev)
enc = PosteriorAudioEncoder(20, 192, 192, 5, 1, 16, gin_channels=16).to(dev)
with torch.no_grad():
    for k, p in enc.named_parameters():
        p.copy_(vqvae_ref.det_fill(k, p.shape, 0.4).to(dev))
res = {}
for n in ("0", "3", "3", "0"):
    os.environ["TTTS_BRANCH_STREAMS"] = n
    enc.zero_grad()
    spec = D("pe_spec").requires_grad_(True); wav = D("pe_wav").requires_grad_(True); gg = D("pe_g").requires_grad_(True)
    z, m, logs = enc(spec, wav, D("pe_mask"), g=gg, noise=D("pe_noise"))
    ct = D("pe_ct")
    ((z * ct).sum() + 0.1 * (m * ct).sum() + 0.1 * logs.sum()).backward()
    torch.cuda.synchronize()
    cur = {"z": z.detach().clone(), "dwav": wav.grad.clone(), "dspec": spec.grad.clone(), "dg": gg.grad.clone()}
    cur.update({"p:" + k: p.grad.clone() for k, p in enc.named_parameters()})
    if n == "0" and "0" not in res:
        res["0"] = cur
        continue
    bad = []
    for k in cur:
        a, b = res["0"][k], cur[k]
        e = ((a - b).abs().max() / (a.abs().max() + 1e-30)).item()
        if e > 1e-4:
            bad.append((k, "%.2e" % e))
    print("streams", n, "differing:", len(bad), bad[:40], flush=True)
