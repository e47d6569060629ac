"""Flat-arena AdamW for ordinary `nn.Module`s: the optimizer of the VQ-VAE-GAN step (torch.optim.AdamW(lr 1e-4,
betas (0.8, 0.99), eps 1e-9) -- ttts/vqvae/train.py:193-205) plus the gradient-norm loop of
commons.clip_grad_value_(params, None) (ttts/utils/commons.py:148-163; 1566 `.item()` syncs per step in the
reference) as THREE kernel launches on one flat fp32 arena.

On construction every trainable parameter is re-homed into a single flat buffer (`p.data` becomes a view of it) and
gets a persistent `.grad` view of a flat gradient buffer, so autograd accumulates straight into the arena, the
data-parallel exchange is one all-reduce of `flat_g`, and `step()` is schedule + norm + AdamW on raw pointers.
Do not call `zero_grad(set_to_none=True)` on the module or move it after wrapping.

Semantic difference from torch.optim.AdamW, documented: a parameter that received no gradient in a step is treated as
having a zero gradient (torch skips it, i.e. applies no weight decay); every parameter of both networks receives a
gradient in the reference step (tests/golden/vqvae_step.npz: 0 unused), so the two coincide on the path.
"""
import torch

from . import ops


class FlatAdamW:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdamW: no trainable parameters")
        dev = self.params[0].device
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4                  # keep every tensor 16-byte aligned
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        for p, o in zip(self.params, self.offsets):
            view = self.flat_p[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        self.state = torch.zeros(8, dtype=torch.float32, device=dev)   # {step, lr, bc1, bc2_sqrt, grad_norm, clip_coef,..}
        self._ws = ops.gradnorm_workspace(off, dev)

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()

    def step(self, lr=None, max_norm=0.0):
        """One AdamW step at learning rate `lr` (default: the constructor's); also measures ||g||_2 (state[4]) and
        leaves the gradient arena zeroed."""
        b1, b2 = self.betas
        ops.adamw_schedule(self.state, self.lr if lr is None else lr, b1, b2, 0)
        ops.gradnorm(self.flat_g, max_norm, self.state, self._ws)
        ops.adamw(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, None, self.state, b1, b2, self.eps,
                  self.weight_decay, zero_grad=True)

    def grad_norm(self):
        """Device scalar (no sync): the gradient 2-norm measured by the last step()."""
        return self.state[4]

    # ---- torch.optim-format state for checkpoints ({'state': {i: {...}}, 'param_groups': [...]}) ---------------------
    def state_dict(self):
        st = {}
        step = self.state[0].item()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            st[i] = {"step": torch.tensor(float(step)), "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                     "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": st, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        step = 0.0
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if i in sd["state"]:
                s = sd["state"][i]
                n = p.numel()
                self.exp_avg[o:o + n].view(p.shape).copy_(s["exp_avg"])
                self.exp_avg_sq[o:o + n].view(p.shape).copy_(s["exp_avg_sq"])
                step = float(s["step"])
        self.state[0] = step
