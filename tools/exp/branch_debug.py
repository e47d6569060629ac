"""Which piece breaks under run_branches?  Gradients of small graphs with TTTS_BRANCH_STREAMS=3 against =0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from ttts_amd import ops
from ttts_amd.vqvae import modules as M

dev = torch.device("cuda", 0)
ops.set_conv_precision(os.environ.get("PREC", "exact"))
torch.manual_seed(0)


def run(name, build, nb):
    torch.manual_seed(1)
    mods = build()
    x0 = torch.randn(4, 32, 2048, device=dev)
    res = {}
    for n in ("0", "3"):
        os.environ["TTTS_BRANCH_STREAMS"] = n
        for m in mods:
            m.zero_grad()
        x = x0.clone().requires_grad_(True)
        h = x * 1.0
        for rep in range(nb):
            xs = M.run_branches([lambda m=m: m(h) for m in mods], dev)
            h = M.add_scale(xs, 1.0 / len(mods))
        (h * h).sum().backward()
        torch.cuda.synchronize()
        res[n] = (x.grad.clone(), [p.grad.clone() for m in mods for p in m.parameters()])
    gx = (res["0"][0] - res["3"][0]).abs().max().item() / res["0"][0].abs().max().item()
    gp = max(((a - b).abs().max() / (a.abs().max() + 1e-30)).item() for a, b in zip(res["0"][1], res["3"][1]))
    print("%-28s rel err dx %.2e  params %.2e" % (name, gx, gp), flush=True)


class TorchConv(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c = torch.nn.Conv1d(32, 32, 3, padding=1)
    def forward(self, x):
        return self.c(torch.nn.functional.leaky_relu(x, 0.1)) + x


for it in range(2):
    run("torch convs", lambda: [TorchConv().to(dev) for _ in range(3)], 2)
    run("Conv1d plain", lambda: [M.Conv1d(32, 32, 3, padding=1).to(dev) for _ in range(3)], 2)
    run("Conv1d weight-norm", lambda: [M._wn_conv_new(32, 3, 1).to(dev) for _ in range(3)], 2)
    run("ResBlock1", lambda: [M.ResBlock1(32, k, (1, 3, 5)).to(dev) for k in (3, 7, 11)], 2)
