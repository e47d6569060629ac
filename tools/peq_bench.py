"""Time the parametric-equaliser augmentation at the BASELINE clip size (32 x 163 840 samples @ 32 kHz, n_fft 2048, hop 640).
Algorithmic HBM bytes per sample: wav read 4 + frames write/read 2 * 4 * n_fft/hop + out write 4 + normalise read/write 8."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ttts_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B, T, N, HOP = int(os.environ.get("PB_B", 32)), 163840, 2048, 640
wav = (torch.rand(B, T, device=dev) - 0.5)
win = torch.hann_window(N, device=dev)
freq = torch.full((B, 10), 1000.0, device=dev); gain = torch.rand(B, 10, device=dev) * 24 - 12; q = torch.rand(B, 10, device=dev) * 3 + 2
kind = torch.tensor([0] * 8 + [1, 2], dtype=torch.int32, device=dev)


def run():
    H = ops.peq_response(freq, gain, q, kind, N, 32000)
    return ops.stft_filter_istft(wav, win, N, HOP, H)


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
bytes_alg = B * T * (4 + 2 * 4 * N / HOP + 4 + 8)
print(json.dumps({"peq_augment_us": round(us, 1), "samples_per_s": round(B * T / us * 1e6), "algorithmic_GB": round(bytes_alg / 1e9, 4),
                  "GBps": round(bytes_alg / us / 1e3, 1), "frac_of_8TBps": round(bytes_alg / us / 1e3 / 8000, 3)}))
