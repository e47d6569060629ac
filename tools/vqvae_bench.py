"""VQ-VAE-GAN step benchmark: B x 163 840-sample clips, full two-phase step, fp32.
Prints one JSON line: spectrogram frames/s = B * 256 / step time.  usage: python tools/vqvae_bench.py [B] [steps] [warmup]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as ge

ge.build()
from ttts_amd.vqvae.train import SyntheticVqvaeBatches, VqvaeTrainer, get_hparams

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 2
hps = get_hparams()
tr = VqvaeTrainer(hps)
cb = tr.net_g.quantizer.vq.layers[0]._codebook
with torch.no_grad():          # codebook pre-initialised (k-means excluded from timing)
    cb.inited.fill_(1); cb.embed.normal_(0, 0.3); cb.embed_avg.copy_(cb.embed * 4); cb.cluster_size.fill_(4.0)
loader = iter(SyntheticVqvaeBatches(B, device=tr.device))
data = next(loader)
for _ in range(warmup):
    out = tr.train_step(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
issue = 0.0
for _ in range(steps):
    ti = time.perf_counter()
    out = tr.train_step(data)
    issue += time.perf_counter() - ti
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# host side alone: the same step with the device parked (every launch only queues), i.e. the time Python + ctypes + autograd need
# to ISSUE a step -- when this approaches ms_per_step the step is host-bound and faster kernels no longer show
if os.environ.get("TTTS_BENCH_HOST", "0") == "1" and hasattr(torch.cuda, "_sleep"):
    torch.cuda.synchronize()
    torch.cuda._sleep(int(4e8))
    ti = time.perf_counter()
    tr.train_step(data)
    host_ms = (time.perf_counter() - ti) * 1e3
    torch.cuda.synchronize()
else:
    host_ms = None
frames = B * (163840 // 640)
print(json.dumps({"metric": "vqvae_gan_train_frames_per_s", "value": frames / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "host_issue_ms_device_parked": host_ms, "host_issue_ms_in_loop": issue / steps * 1e3,
                  "batch": B, "steps": steps, "warmup": warmup, "dtype": "f32 (conv products split-bf16 hi/lo x3 on the bf16 MFMA, fp32 accumulate; TTTS_CONV_FP32=1 for exact fp32 MFMA)", "data": "synthetic",
                  "tflops_algorithmic": 1.97e9 * frames / dt / 1e12, "max_mem_gb": torch.cuda.max_memory_allocated() / 2**30,
                  "launch_batching": {"wsplit": [c.stats() for c in tr.step_fn.wsplit_g + tr.step_fn.wsplit_d],
                                      "slabs": [a.stats() for a in tr.step_fn.slabs_g + tr.step_fn.slabs_d]},
                  "losses": {k: float(v) for k, v in out.items()}}))
