"""Single-process recorded VQ-VAE-GAN step with a given set of side-stream pools allowed under capture."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from ttts_amd.vqvae.train import SyntheticVqvaeBatches, VqvaeTrainer, get_hparams
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
hps = get_hparams()
tr = VqvaeTrainer(hps)
cb = tr.net_g.quantizer.vq.layers[0]._codebook
with torch.no_grad():
    cb.inited.fill_(1); cb.embed.normal_(0, 0.3); cb.embed_avg.copy_(cb.embed * 4); cb.cluster_size.fill_(4.0)
data = next(iter(SyntheticVqvaeBatches(B, device=tr.device)))
for _ in range(3):
    out = tr.train_step_graphed(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = tr.train_step_graphed(data)
torch.cuda.synchronize()
print("CAPTURE-OK pools=%s B=%d ms=%.2f loss=%.4f" % (os.environ.get("TTTS_CAPTURE_POOLS"), B, (time.perf_counter() - t0) / 5 * 1e3, float(out["loss_gen_all"])), flush=True)
