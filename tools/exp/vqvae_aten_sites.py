"""Where do the PyTorch elementwise launches of a VQ-VAE-GAN step come from?  One eager step under a TorchDispatchMode (autograd
single-threaded so that the backward is seen too): aten fill_ / add_ / add / copy_ ... grouped by the innermost ttts_amd frame."""
import collections, sys, os, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ttts_amd.vqvae.train import SyntheticVqvaeBatches, VqvaeTrainer, get_hparams

hps = get_hparams()
tr = VqvaeTrainer(hps)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
loader = iter(SyntheticVqvaeBatches(B, seed=3, device=tr.device))
for _ in range(2):
    tr.train_step(next(loader))
torch.cuda.synchronize()
WATCH = ("fill_", "zero_", "add_", "add.", "copy_", "mul.", "cat", "zeros", "clone", "sub.", "div.")
sites = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(w in name for w in WATCH):
            big = any(torch.is_tensor(a) and a.is_cuda for a in args)
            if big:
                frames = [f for f in traceback.extract_stack() if "ttts_amd" in f.filename]
                where = "%s:%d %s" % (frames[-1].filename.split("ttts_amd/")[-1], frames[-1].lineno, frames[-1].name) if frames else "(autograd engine / torch)"
                if not frames:
                    fr = traceback.extract_stack()
                    where += " <- " + " / ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-6:-2])
                sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)
batch = next(loader)
with Rec():
    tr.train_step(batch)
torch.cuda.synchronize()
tot = collections.Counter()
for (n, w), c in sites.items():
    tot[n] += c
print(tot.most_common(12))
for (n, w), c in sites.most_common(40):
    print("%5d  %-28s %s" % (c, n, w[:170]))
