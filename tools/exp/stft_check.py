"""stft_mag against a float64 torch.stft on the same clips: max abs error / max |spec| and relative L2, for the default kernel and
(TTTS_STFT_R4=1) the radix-4 kernel."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge; ge.build()
from ttts_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
wav = (torch.rand(6, 163840, generator=g) - 0.5)
win = torch.hann_window(2048, dtype=torch.float64)
pad = (2048 - 640) // 2
xp = torch.nn.functional.pad(wav.double().unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
ref = torch.stft(xp, 2048, 640, 2048, win, center=False, return_complex=True)
ref = torch.sqrt(ref.real ** 2 + ref.imag ** 2 + 1e-6)
s = ops.stft_mag(wav.to(dev), win.float().to(dev), 2048, 640).double().cpu()
err = (s - ref).abs()
print("kernel %s: shape %s max abs err %.3e (max |spec| %.3f) rel L2 %.3e worst bin %d frame %d" % (
    os.environ.get("TTTS_STFT_R4", "w32"), tuple(s.shape), float(err.max()), float(ref.max()),
    float((s - ref).norm() / ref.norm()), int(err.amax(dim=(0, 2)).argmax()), int(err.amax(dim=(0, 1)).argmax())))
