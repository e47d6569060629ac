"""Flat-arena AdamW for ordinary `nn.Module`s: the optimizer of the VQ-VAE-GAN step (torch.optim.AdamW(lr 1e-4,
betas (0.8, 0.99), eps 1e-9) -- ttts/vqvae/train.py:193-205) plus the gradient-norm loop of
commons.clip_grad_value_(params, None) (ttts/utils/commons.py:148-163; 1566 `.item()` syncs per step in the
reference) as THREE kernel launches on one flat fp32 arena.

On construction every trainable parameter is re-homed into a single flat buffer (`p.data` becomes a view of it) and
gets a persistent `.grad` view of a flat gradient buffer, so autograd accumulates straight into the arena, the
data-parallel exchange is one all-reduce of `flat_g`, and `step()` is schedule + norm + AdamW on raw pointers.
`optimizer.zero_grad()` keeps the views whatever `set_to_none` says; a stray `module.zero_grad(set_to_none=True)` drops
them, `step()` / `attach_grads()` re-attaches them (the arena itself is zeroed by every step).  Do not move the module
after wrapping.

Semantic difference from torch.optim.AdamW, documented: a parameter that received no gradient in a step is treated as
having a zero gradient (torch skips it, i.e. applies no weight decay); every parameter of both networks receives a
gradient in the reference step (tests/golden/vqvae_step.npz: 0 unused), so the two coincide on the path.
"""
import torch

from . import ops


class FlatAdamW(torch.optim.Optimizer):
    """A `torch.optim.Optimizer` (so `ExponentialLR(optim, ...)`, `scaler.step(optim)` and `optim.param_groups[0]["lr"]`
    of the reference's `run` / `train_and_evaluate` work unchanged, ttts/vqvae/train.py:193-205,281-295,365-401) whose
    `step()` is three kernel launches over one flat arena."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FlatAdamW: no trainable parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False))
        self.params = self.param_groups[0]["params"]
        dev = self.params[0].device
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
        self.opt_state = torch.zeros(8, dtype=torch.float32, device=dev)   # {step, lr, bc1, bc2_sqrt, grad_norm, clip_coef,..}
        self._ws = ops.gradnorm_workspace(off, dev)

    # hyper-parameters live in param_groups[0] (what LR schedulers edit)
    lr = property(lambda self: self.param_groups[0]["lr"], lambda self, v: self.param_groups[0].__setitem__("lr", v))
    betas = property(lambda self: tuple(self.param_groups[0]["betas"]))
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    def zero_grad(self, set_to_none=False):
        """Zeroes the gradient arena; the `.grad` views stay attached whatever `set_to_none` says."""
        self.flat_g.zero_()

    def attach_grads(self):
        """Re-attach the `.grad` views (after a stray `module.zero_grad(set_to_none=True)` dropped them)."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_g[o:].data_ptr():
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def step(self, closure=None, lr=None, max_norm=0.0, warmup_steps=0):
        """One AdamW step at learning rate `lr` (default: param_groups[0]["lr"]); also measures ||g||_2 (opt_state[4])
        and leaves the gradient arena zeroed.  warmup_steps > 0: `lr` is the BASE rate and the LambdaLR warm-up factor
        min(1, step / warmup_steps) is applied on the device from the optimizer's own step counter -- a recorded (hipGraph) step
        cannot take a new host-side rate every replay."""
        if closure is not None:
            raise NotImplementedError("FlatAdamW.step: closures are not part of the path")
        self.attach_grads()
        b1, b2 = self.betas
        ops.adamw_schedule(self.opt_state, self.lr if lr is None else lr, b1, b2, int(warmup_steps))
        ops.gradnorm(self.flat_g, max_norm, self.opt_state, self._ws)
        ops.adamw(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, None, self.opt_state, b1, b2, self.eps,
                  self.weight_decay, zero_grad=True)

    def measure_grad_norm(self):
        """||g||_2 of the arena as it is NOW (device scalar, no sync): commons.clip_grad_value_(params, None)."""
        ops.gradnorm(self.flat_g, 0.0, self.opt_state, self._ws)
        return self.opt_state[4].clone()

    def grad_norm(self):
        """Device scalar (no sync): the gradient 2-norm measured by the last step()."""
        return self.opt_state[4]

    # ---- torch.optim-format state for checkpoints ({'state': {i: {...}}, 'param_groups': [...]}) ---------------------
    def state_dict(self):
        st = {}
        step = self.opt_state[0].item()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            st[i] = {"step": torch.tensor(float(step)), "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                     "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["betas"] = tuple(group["betas"])
        group["params"] = list(range(len(self.params)))
        return {"state": st, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            self.param_groups[0][k] = tuple(g[k]) if k == "betas" else g[k]
        if "initial_lr" in g:
            self.param_groups[0]["initial_lr"] = g["initial_lr"]
        step = 0.0
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if i in sd["state"]:
                s = sd["state"][i]
                n = p.numel()
                self.exp_avg[o:o + n].view(p.shape).copy_(s["exp_avg"])
                self.exp_avg_sq[o:o + n].view(p.shape).copy_(s["exp_avg_sq"])
                step = float(s["step"])
        self.opt_state[0] = step
