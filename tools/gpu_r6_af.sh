#!/bin/bash
# the wave-per-frame STFT kernel: parity + timing against the radix-4 kernel (TTTS_STFT_R4=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -x -k "stft or mel" 2>&1 | tail -3
cat > /tmp/stft_t.py <<'P'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.build()
from ttts_amd import ops
dev = torch.device("cuda", 0)
wav = (torch.rand(32, 163840) - 0.5).to(dev); win = torch.hann_window(2048).to(dev)
for _ in range(3): s = ops.stft_mag(wav, win, 2048, 640)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): s = ops.stft_mag(wav, win, 2048, 640)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); 
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
ref = torch.stft(wav, 2048, 640, 2048, win, center=False, pad_mode="reflect", normalized=False, onesided=True, return_complex=True) if False else None
print("stft_mag %s: %.2f us per call, checksum %.6f" % (os.environ.get("TTTS_STFT_R4", "w32"), e0.elapsed_time(e1) * 1e3 / 100, float(s.double().sum())))
P
python /tmp/stft_t.py 2>&1 | tail -1
TTTS_STFT_R4=1 python /tmp/stft_t.py 2>&1 | tail -1
