"""Trainer of the diffusion mel-denoiser on the HIP kernels: the step body of ttts/diffusion/train.py:156-200 with the same
recipe (AdamW(lr, (0.9, 0.999), wd 0.01) :119, LambdaLR warm-up over 1000 steps :69-73,120, clip 1.0 :195, checkpoint dict
`{'step', 'model'}` :135-141) and config keys (ttts/diffusion/config.yaml: train.*, aa_diffusion.*).

Host-side differences (outside the arithmetic): one process per GPU under torchrun with one flat gradient all-reduce
(`parallel.FlatDataParallel`) instead of accelerate; `FlatAdamW` (one arena, fused grad-norm / clip / AdamW, no per-tensor
`.item()`); the GPT latent is an input of `train_step` (the frozen GPT lives in `ttts_amd.gpt`, `return_latent=True`); data
loading, EMA copy, vocoder previews and tensorboard are outside the path.
"""
import os

import torch

from .. import ops

from ..optim import FlatAdamW
from ..parallel import FlatDataParallel, init_distributed
from . import aa_model as _aa
from .aa_model import AA_diffusion, normalize_tacotron_mel
from .gaussian import SpacedDiffusion, get_named_beta_schedule, space_timesteps


def warmup(step):
    """train.py:69-73."""
    return float(step / 1000) if step < 1000 else 1


class DiffusionTrainer:
    def __init__(self, cfg, device=None, seed=0):
        """cfg: dict with the keys of ttts/diffusion/config.yaml (`train.lr`, `train.timesteps`, `aa_diffusion.*`)."""
        self.rank, self.world, local = init_distributed()
        self.device = torch.device("cuda", local) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ttts_amd.diffusion.train needs a GPU (no CPU fallback)")
        torch.cuda.set_device(self.device)
        self.cfg = cfg
        torch.manual_seed(seed)
        self.diffusion = AA_diffusion(**cfg["aa_diffusion"]).to(self.device)
        steps = int(cfg["train"].get("timesteps", 1000))
        self.diffuser = SpacedDiffusion(use_timesteps=space_timesteps(steps, [steps]), model_mean_type="epsilon",
                                        model_var_type="learned_range", loss_type="mse",
                                        betas=get_named_beta_schedule("linear", steps), conditioning_free=False, conditioning_free_k=2.0)
        self.desired_diffusion_steps = steps
        self.dp = FlatDataParallel()
        self.optimizer = FlatAdamW(self.diffusion.parameters(), cfg["train"]["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
        self.dp.broadcast_(self.optimizer.flat_p)
        # t and the noise are drawn per rank (accelerate leaves every process its own RNG stream): a shared seed would
        # put the same timesteps and noise on every replica
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) + 7919 * self.dp.rank + 1)
        self.base_lr = cfg["train"]["lr"]
        self.step = 0
        self.diffusion.train()
        # launch batching shared with the VQ-VAE-GAN step (ops.WeightSplitCache / ops.WgradSlabArena): the bf16 hi/lo copies of
        # every convolution weight rewritten by one launch per step, all split-K weight-gradient slabs summed by one launch
        self._wsplit, self._slabs = [], []
        try:
            if os.environ.get("TTTS_WSPLIT_CACHE", "1") == "1":
                self._wsplit = [ops.WeightSplitCache(self.optimizer.flat_p)]
            if os.environ.get("TTTS_WGRAD_ARENA", "1") == "1":
                self._slabs = [ops.WgradSlabArena(self.optimizer.flat_g, int(os.environ.get("TTTS_DIFFUSION_ARENA_MB", "2048")) << 20)]
        except ops.TttsError:           # (a second trainer over the same arrays: it runs without them)
            pass

    # ---- the step as hipGraph replays ------------------------------------------------------------------------------------------
    def train_step_graphed(self, mel, mel_refer, latent, normalized=False, max_graphs=24, max_dropped=1):
        """The same step replayed from a recorded hipGraph (the eager step is ~2 700 launches issued from Python: with the fused
        attention kernels its device time is well below the host's issue time).  The forward's one HOST-side random choice --
        which layers `layer_drop` skips (aa_model.py:268-277: random.random() per layer) -- is drawn here, exactly as the eager
        forward draws it, and selects WHICH recording is replayed: one graph per drop pattern that has occurred (no drop and the
        seven single drops cover 85 % of the steps at layer_drop 0.1), all sharing one memory pool.  Patterns with at most
        `max_dropped` skipped layers are recorded up front, on the first call with a new shape (recording does not run the step);
        rarer patterns run launch by launch -- a recording costs several eager steps and would be replayed once in a hundred.  t and the noise are drawn into static buffers in front of the replay (the trainer's own
        generator); the unconditioned mask is drawn inside the graph (torch's capture-aware default generator); the warm-up
        factor of the learning rate is applied on the device from the optimizer's step counter.  Needs two eager steps first
        (lazy initialisation of caches and arenas).  The returned scalars are the graph's static outputs: read them before the
        next call."""
        import random
        model = self.diffusion
        n = len(model.layers)
        drop = tuple(i for i in range(n)
                     if model.training and model.layer_drop > 0 and i != 0 and i != n - 1 and random.random() < model.layer_drop)
        if self.step < 2 or self.dp.enabled:
            return self.train_step(mel, mel_refer, latent, inject={"drop_layers": drop}, normalized=normalized)
        key = (tuple(mel.shape), tuple(mel_refer.shape), tuple(latent.shape), bool(normalized), ops.conv_precision(),
               _aa._PRECISION["mode"])
        st = getattr(self, "_gstate", None)
        if st is None or st["key"] != key:
            for a in self._slabs:
                a.release_graphs()
            st = self._gstate = {"key": key, "graphs": {}, "pool": None, "failed": False,
                                 "mel": mel.clone(), "ref": mel_refer.clone(), "lat": latent.clone(),
                                 "t": torch.zeros(mel.shape[0], dtype=torch.long, device=self.device),
                                 "noise": torch.zeros_like(mel)}
        st["mel"].copy_(mel); st["ref"].copy_(mel_refer); st["lat"].copy_(latent)
        st["t"].copy_(torch.randint(0, self.desired_diffusion_steps, (mel.shape[0],), device=self.device, generator=self.gen))
        st["noise"].copy_(torch.randn(mel.shape, device=self.device, dtype=mel.dtype, generator=self.gen))
        def record(pattern):
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                alive = torch.distributed.is_available() and torch.distributed.is_initialized()
                with torch.cuda.graph(g, pool=st["pool"], capture_error_mode="thread_local" if alive else "global"):
                    out = self._step_body(st["mel"], st["ref"], st["lat"], st["t"], st["noise"], {"drop_layers": pattern}, normalized,
                                          device_warmup=True)
                if st["pool"] is None:
                    st["pool"] = g.pool()
                st["graphs"][pattern] = (g, out)              # (capture only records: the step itself still has to run)
            except Exception as err:                         # noqa: BLE001 -- refused: say so once, run launch by launch from now on
                import sys
                print("ttts_amd: hipGraph capture of the diffusion step failed (%s); running it launch by launch"
                      % str(err).splitlines()[0][:200], file=sys.stderr, flush=True)
                st["failed"] = True
                torch.cuda.synchronize()
        if not st["graphs"] and not st["failed"]:            # first call with this shape: the frequent patterns, up front
            import itertools
            droppable = [i for i in range(1, n - 1)] if (model.training and model.layer_drop > 0) else []
            for k in range(0, max_dropped + 1):
                for pattern in itertools.combinations(droppable, k):
                    if len(st["graphs"]) < max_graphs and not st["failed"]:
                        record(tuple(pattern))
        ent = st["graphs"].get(drop)
        if ent is None and not st["failed"] and len(drop) <= max_dropped and len(st["graphs"]) < max_graphs:
            record(drop)
            ent = st["graphs"].get(drop)
        if ent is None:
            return self._step_body(st["mel"], st["ref"], st["lat"], st["t"], st["noise"], {"drop_layers": drop}, normalized,
                                   device_warmup=False)
        ent[0].replay()
        self.step += 1
        return ent[1]

    def train_step(self, mel, mel_refer, latent, t=None, noise=None, inject=None, normalized=False):
        """mel (B, 100, T) / mel_refer (B, 100, Tr) raw log-mels (`normalized=True`: already through normalize_tacotron_mel),
        latent (B, 512, T / 4) GPT latents (already transposed, train.py:161-165).  Returns {"loss", "grad_norm"} device scalars."""
        return self._step_body(mel, mel_refer, latent, t, noise, inject, normalized, device_warmup=False)

    def _step_body(self, mel, mel_refer, latent, t, noise, inject, normalized, device_warmup):
        x_start = mel if normalized else normalize_tacotron_mel(mel)
        refer = mel_refer if normalized else normalize_tacotron_mel(mel_refer)
        if t is None:
            t = torch.randint(0, self.desired_diffusion_steps, (x_start.shape[0],), device=self.device, generator=self.gen)
        if noise is None:
            noise = torch.randn(x_start.shape, device=self.device, dtype=x_start.dtype, generator=self.gen)
        kw = {"latent": latent, "refer": refer}
        kw.update(inject or {})
        try:
            for c in self._wsplit:
                c.refresh()
            out = self.diffuser.training_losses(self.diffusion, x_start, t, model_kwargs=kw, noise=noise)
            loss = out["loss_mean"]
            self.optimizer.zero_grad()
            for a in self._slabs:
                a.begin()
            # ('tf32class' convolutions round the data gradient's input to fp16: a power-of-two loss scale keeps it in range, see
            # ttts_amd/vqvae/train.py; divided out of the arena below)
            # the scale is dynamic: ops.DynamicLossScale, GradScaler's halve / skip / grow rule on device words)
            lsc = None
            if ops.conv_precision() == "tf32class":
                if getattr(self, "_lsc", None) is None:
                    self._lsc = ops.DynamicLossScale.from_env(self.device)
                lsc = self._lsc
            if lsc is None:
                (loss * self.dp.loss_scale()).backward()
            else:
                ((loss * self.dp.loss_scale()) * lsc.scale).backward()
            for a in self._slabs:
                a.reduce()
            if lsc is not None:
                lsc.fetch()
        finally:
            for c in self._wsplit + self._slabs:
                c.disarm()
        if lsc is not None and self.dp.enabled:
            torch.distributed.all_reduce(lsc.events, group=self.dp.group)
        self.dp.allreduce_grads_(self.optimizer.flat_g)
        # (detached: a returned tensor that still carries its grad_fn keeps this step's autograd graph -- and with it the parameters'
        # AccumulateGrad nodes, bound to the stream they were created on -- alive in the caller's hands; a later hipGraph capture
        # then inherits nodes of the default stream and hipStreamEndCapture crashes)
        res = {"loss": loss.detach(), "terms": {k: v.detach() for k, v in out.items()}}
        if lsc is not None:
            lsc.decide(self.optimizer.opt_state)
            self.optimizer.flat_g.mul_(lsc.inv_scale)
            res.update({"loss_scale": lsc.scale.clone(), "f16_saturated": lsc.saturated, "f16_flushed": lsc.flushed,
                        "skipped_steps": lsc.skipped})
        if device_warmup:       # recorded step: the factor comes from the optimizer's device-side step counter (same formula)
            self.optimizer.step(lr=self.base_lr, max_norm=1.0, warmup_steps=1000)
        else:
            lr = self.base_lr * warmup(self.step)                         # LambdaLR: the factor of the step being taken
            self.optimizer.step(lr=lr, max_norm=1.0)
        if lsc is not None:
            lsc.update()
        if not (device_warmup and torch.cuda.is_current_stream_capturing()):
            self.step += 1
        res["grad_norm"] = self.optimizer.grad_norm()
        return res

    def save(self, path):
        if self.rank == 0:
            torch.save({"step": self.step, "model": self.diffusion.state_dict()}, path)

    def load(self, path):
        data = torch.load(path, map_location=self.device)
        self.step = data["step"]
        self.diffusion.load_state_dict(data["model"], strict=False)
