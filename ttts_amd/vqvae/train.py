"""VQ-VAE-GAN trainer on the HIP kernels: the step body of ttts/vqvae/train.py:313-406 (`train_and_evaluate`), the
setup of `run` (:119-297) and `main` (:44-60), with the same config keys (ttts/vqvae/config.json) and checkpoint layout
(`{'model','iteration','optimizer','learning_rate'}`, files `G_{step}.pth` / `D_{step}.pth`).

Host-side differences (all outside the arithmetic):
  * one process per GPU under torchrun; the two DDP wrappers become two flat all-reduces (after the D backward and after
    the G backward) through `parallel.FlatDataParallel`; the rank-0 codebook broadcast of DDP's buffer sync is explicit;
  * `FlatAdamW` replaces the two AdamW instances and the 1566 `.item()` calls of `clip_grad_value_(…, None)`;
  * losses stay on the device until `log_interval`;
  * data: `dataset.path == "synthetic"` feeds band-limited noise clips of the collater's dict shape;
    `wav_aug = wav` (the reference's freeze_quantizer branch) unless the trainer is built with `use_augment=True`
    (config key `train.augment`), which runs the parametric-equaliser part of the reference's augmentation on the GPU
    (`augment.py`; the Praat formant / pitch stage is a CPU library and stays outside).
"""
import glob
import json
import os
import re
import sys

import torch

from .. import ops
from ..optim import FlatAdamW
from ..parallel import FlatDataParallel, init_distributed
from ..utils.data_utils import HParams, mel_spectrogram_torch, spec_to_mel_torch, spectrogram_torch
from . import losses as L
from .augment import Augment, augment, sample_like  # noqa: F401
from .modules import join_side_streams
from .vq2 import MultiPeriodDiscriminator, SynthesizerTrn, slice_segments

global_step = 0


def get_hparams(config_path=os.path.join(os.path.dirname(__file__), "config.json")):
    return HParams(**json.load(open(config_path)))


# ---- checkpoints (ttts/utils/vc_utils.py:248-330) ----------------------------------------------------------------------
def save_checkpoint(model, optimizer, learning_rate, iteration, checkpoint_path):
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({"model": state, "iteration": iteration, "optimizer": optimizer.state_dict(),
                "learning_rate": learning_rate}, checkpoint_path)


def load_checkpoint(checkpoint_path, model, optimizer=None, skip_optimizer=False):
    ck = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    own = model.state_dict()
    new = {}
    for k, v in own.items():                      # tolerant load, like the reference: keep own tensor on mismatch
        new[k] = ck["model"][k] if k in ck["model"] and ck["model"][k].shape == v.shape else v
    model.load_state_dict(new)
    if optimizer is not None and not skip_optimizer and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    return model, optimizer, ck["learning_rate"], ck["iteration"]


def latest_checkpoint_path(dir_path, regex="G_*.pth"):
    files = glob.glob(os.path.join(dir_path, regex))
    files.sort(key=lambda f: int("".join(filter(str.isdigit, os.path.basename(f))) or 0))
    return files[-1]


class SyntheticVqvaeBatches:
    """Endless seeded batches with the keys of VQVAECollater: band-limited noise in [-1, 1] (16 sinusoids + 0.05 N(0,1))."""

    def __init__(self, batch_size, n_samples=163840, text_len=64, seed=1234, device="cpu"):
        self.B, self.n, self.tl, self.device = batch_size, n_samples, text_len, device
        self.g = torch.Generator().manual_seed(seed)

    def __iter__(self):
        return self

    def __next__(self):
        t = torch.arange(self.n, dtype=torch.float32) / 32000.0
        f = torch.rand(self.B, 16, 1, generator=self.g) * 6000 + 60
        a = torch.rand(self.B, 16, 1, generator=self.g) * 0.18 + 0.02
        ph = torch.rand(self.B, 16, 1, generator=self.g) * 6.2832
        wav = (a * torch.sin(6.2832 * f * t.view(1, 1, -1) + ph)).sum(1) + 0.05 * torch.randn(self.B, self.n, generator=self.g)
        return {"wav": wav.clamp(-1, 1).to(self.device), "wav_lengths": torch.full((self.B,), self.n, dtype=torch.int64, device=self.device),
                "text": torch.randint(1, 255, (self.B, self.tl), generator=self.g).to(self.device),
                "text_lengths": torch.full((self.B,), self.tl, dtype=torch.int64, device=self.device)}


class VqvaeStep:
    """The two-phase step body of `train_and_evaluate` (ttts/vqvae/train.py:313-406) over caller-owned parts:
    `nets = [net_g, net_d]`, `optims = [optim_g, optim_d]` (FlatAdamW), `aug` (Augment or None)."""

    def __init__(self, hps, net_g, net_d, optim_g, optim_d, aug=None, dp=None):
        self.hps, self.net_g, self.net_d, self.optim_g, self.optim_d, self.aug = hps, net_g, net_d, optim_g, optim_d, aug
        self.dp = dp if dp is not None else FlatDataParallel()
        self._d_params = [prm for prm in net_d.parameters() if prm.requires_grad]
        # every weight-normed layer of a network is (re)normalised by one launch per phase (modules.WeightNormBank); TTTS_WN_BANK=0
        # restores one launch per convolution call
        self.bank_g = self.bank_d = None
        if os.environ.get("TTTS_WN_BANK", "1") == "1":
            from .modules import WeightNormBank
            self.bank_g, self.bank_d = WeightNormBank(net_g), WeightNormBank(net_d)
        # the bf16 hi/lo operand copies of every convolution weight are rewritten by one launch per phase as well
        # (ops.WeightSplitCache over the banks' effective weights and the generator's parameter arena); TTTS_WSPLIT_CACHE=0: one
        # split launch in front of every convolution call
        self.wsplit_g, self.wsplit_d = [], []
        if self.bank_g is not None and os.environ.get("TTTS_WSPLIT_CACHE", "1") == "1":
            from .. import ops
            def _try(make, t, *a):      # (a second step object over the same arrays -- trainer + train_and_evaluate -- finds the
                if t is None:           # range registered already: it runs without its own cache / arena, results are the same)
                    return []
                try:
                    return [make(t, *a)]
                except ops.TttsError:
                    return []
            self.wsplit_g = _try(ops.WeightSplitCache, getattr(self.bank_g, "flat_w", None)) + _try(ops.WeightSplitCache, optim_g.flat_p)
            self.wsplit_d = _try(ops.WeightSplitCache, getattr(self.bank_d, "flat_w", None))
        # ... and the split-K partial sums of every weight gradient are added into dW / the gradient arena by one launch per
        # backward (ops.WgradSlabArena; TTTS_WGRAD_ARENA=0: a reduce launch behind every weight-gradient call).  Storage in MiB:
        # TTTS_WGRAD_ARENA_MB = "generator bank, generator arena, discriminator bank, discriminator arena"
        self.slabs_g, self.slabs_d = [], []
        if self.bank_g is not None and os.environ.get("TTTS_WGRAD_ARENA", "1") == "1":
            from .. import ops
            mb = [int(v) << 20 for v in os.environ.get("TTTS_WGRAD_ARENA_MB", "8192,2048,1024,64").split(",")]
            def _try_a(t, n):
                if t is None:
                    return []
                try:
                    return [ops.WgradSlabArena(t, n)]
                except ops.TttsError:
                    return []
            self.slabs_g = _try_a(getattr(self.bank_g, "flat_dw", None), mb[0]) + _try_a(optim_g.flat_g, mb[1])
            self.slabs_d = _try_a(getattr(self.bank_d, "flat_dw", None), mb[2]) + _try_a(optim_d.flat_g, mb[3])

    def _sync_buffers(self):
        if self.dp.enabled:                                                  # DDP broadcast_buffers=True (rank-0 codebook)
            self.dp.broadcast_(*[b for b in self.net_g.buffers() if b.is_floating_point()])

    def _exchange(self, which, cut):
        """The step's collectives: 1 = discriminator gradients, 2 = generator gradients (flat SUM all-reduces).  `cut(fn)`, when
        given, is how a caller that records the step as hipGraph segments takes the collective OUT of the recording: it ends
        the current segment, runs fn, starts the next one (VqvaeTrainer._capture)."""
        arena = self.optim_d.flat_g if which == 1 else self.optim_g.flat_g
        lsc = self._loss_scaler()

        def fn():
            if lsc is not None and self.dp.enabled:      # the overflow decision must be the same on every rank: SUM of the event counts
                torch.distributed.all_reduce(lsc.events, group=self.dp.group)
            self.dp.allreduce_grads_(arena)
        if cut is None:
            fn()
        else:
            cut(fn)

    def _loss_scaler(self):
        """The dynamic loss scale of the 'tf32class' convolution mode (ops.DynamicLossScale: GradScaler's rule on device words;
        the reference builds `GradScaler(enabled=hps.train.fp16_run)`, vqvae/train.py:262), or None in the fp32-equivalent modes.
        TTTS_LOSS_SCALE = initial scale (default 2^10), TTTS_LOSS_SCALE_DYNAMIC=0 pins it, TTTS_LOSS_SCALE_INTERVAL = clean steps
        before it doubles (2000)."""
        if ops.conv_precision() != "tf32class":
            return None
        if getattr(self, "_lsc", None) is None:
            self._lsc = ops.DynamicLossScale.from_env(self.optim_g.flat_g.device)
        return self._lsc

    def __call__(self, data, inject=None, cut=None, sync_buffers=True):
        """data: dict(wav (B, T) f32, wav_lengths, text, text_lengths) on the device.  Returns a dict of device scalars."""
        h, tr = self.hps.data, self.hps.train
        inject = dict(inject or {})
        if sync_buffers:
            self._sync_buffers()      # (first thing in the step: nothing before the generator forward touches a buffer)
        wav, wav_lengths, text, text_lengths = data["wav"], data["wav_lengths"], data["text"], data["text_lengths"]
        y = wav
        spec = spectrogram_torch(wav, h.filter_length, h.hop_length, h.win_length, center=False)
        spec_lengths = torch.div(wav_lengths, h.hop_length, rounding_mode="floor")
        if "wav_aug" in inject:
            wav_aug = inject.pop("wav_aug")                                   # tests: a fixed augmented clip
        elif self.aug is None or getattr(self.hps.vqvae, "freeze_quantizer", None) is True:
            wav_aug = wav                                                     # train.py:335-336
        else:
            wav_aug = augment(wav, self.aug, self.hps)                        # train.py:337-338 (PEQ part)
        spec_aug = spec if wav_aug is wav else spectrogram_torch(wav_aug, h.filter_length, h.hop_length, h.win_length, center=False)
        # No host read-back inside the step: the dead-code replacement of the quantizer (core_vq.py:152-168) asks the host whether any
        # code expired -- a device -> host sync in the middle of the generator forward that stalls the launch stream for ~16 ms of a
        # 132 ms step (the host waits for the encoders, then the device waits for the host to catch up).  What it would write never
        # survives the step, in the reference either (quantize.expire_codes_ has the line-by-line argument), so the trainer's step
        # skips it, as the recorded (hipGraph) step always had to.  TTTS_VQ_EXPIRE_SYNC=1 restores the read-back.
        from .quantize import EuclideanCodebook
        prev_sync_free = EuclideanCodebook.sync_free
        if os.environ.get("TTTS_VQ_EXPIRE_SYNC", "0") != "1":
            EuclideanCodebook.sync_free = True
        try:
            return self._phases(wav, wav_aug, wav_lengths, spec, spec_aug, spec_lengths, text, text_lengths, y, inject, cut)
        finally:
            EuclideanCodebook.sync_free = prev_sync_free
            for bank in (self.bank_g, self.bank_d):
                if bank is not None:
                    bank.release()
            for c in self.wsplit_g + self.wsplit_d + self.slabs_g + self.slabs_d:
                c.disarm()

    def _phases(self, wav, wav_aug, wav_lengths, spec, spec_aug, spec_lengths, text, text_lengths, y, inject, cut):
        h, tr = self.hps.data, self.hps.train
        if self.bank_g is not None:
            self.bank_g.refresh()
        for c in self.wsplit_g:
            c.refresh()
        for a in self.slabs_g:
            a.begin()
        y_hat, kl_ssl, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), quantized = self.net_g(
            wav, wav_aug, wav_lengths, spec, spec_aug, spec_lengths, text, text_lengths, **inject)
        mel = spec_to_mel_torch(spec, h.filter_length, h.n_mel_channels, h.sampling_rate, h.mel_fmin, h.mel_fmax)
        y_mel = slice_segments(mel, ids_slice, tr.segment_size // h.hop_length)
        y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1), h.filter_length, h.n_mel_channels, h.sampling_rate, h.hop_length,
                                          h.win_length, h.mel_fmin, h.mel_fmax)
        y = slice_segments(y.unsqueeze(1), ids_slice * h.hop_length, tr.segment_size)
        scale = self.dp.loss_scale()
        # 'tf32class' convolutions round the data gradient's input to fp16: a power-of-two loss scale keeps the GAN's small
        # gradients (1e-7 .. 1e-2 per element) inside fp16's normal range; it is divided out of the flat gradient arenas before the
        # optimizer (exact: a power of two), so the reported gradient norms and the updates are the unscaled ones.  The scale is
        # DYNAMIC (GradScaler's rule, device-side): a saturated conversion in a backward skips that optimizer step and halves the scale
        lsc = self._loss_scaler()
        ls = lsc.scale if lsc is not None else 1.0
        # ---- discriminator phase
        if self.bank_d is not None:
            self.bank_d.refresh()
        for c in self.wsplit_d:
            c.refresh()
        for a in self.slabs_d:
            a.begin()
        y_d_hat_r, y_d_hat_g, _, _ = self.net_d(y, y_hat.detach())
        loss_disc, losses_disc_r, losses_disc_g = L.discriminator_loss(y_d_hat_r, y_d_hat_g)
        self.optim_d.zero_grad()
        (loss_disc * scale if lsc is None else (loss_disc * scale) * ls).backward()
        join_side_streams(y.device)
        for a in self.slabs_d:
            a.reduce()
        if self.bank_d is not None:
            self.bank_d.finish()
        if lsc is not None:
            lsc.fetch()
        self._exchange(1, cut)
        if lsc is not None:
            lsc.decide(self.optim_d.opt_state)
            self.optim_d.flat_g.mul_(lsc.inv_scale)
        self.optim_d.step()
        # ---- generator phase.  The reference lets this backward fill net_d's parameter gradients too and throws them away at
        # the next `optim_d.zero_grad()` (vqvae/train.py:354-372 there): here the discriminator's parameters are frozen for the
        # phase, so autograd only runs the data gradients the generator needs -- same losses, same updates, no discarded
        # weight-gradient / weight-norm / bias-gradient launches.
        for prm in self._d_params:
            prm.requires_grad_(False)
        try:
            if self.bank_d is not None:
                self.bank_d.refresh(requires_grad=False)          # the UPDATED discriminator, frozen
            for c in self.wsplit_d:
                c.refresh()
            y_d_hat_r, y_d_hat_g, fmap_r, fmap_g = self.net_d(y, y_hat)
            loss_mel = L.l1_loss(y_mel, y_hat_mel) * tr.c_mel
            loss_kl = L.kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * tr.c_kl
            loss_fm = L.feature_loss(fmap_r, fmap_g)
            loss_gen, losses_gen = L.generator_loss(y_d_hat_g)
            loss_gen_all = loss_gen + loss_fm + loss_mel + kl_ssl * 1 + loss_kl
            self.optim_g.zero_grad()
            (loss_gen_all * scale if lsc is None else (loss_gen_all * scale) * ls).backward()
            join_side_streams(y.device)
            for a in self.slabs_g:
                a.reduce()
            if self.bank_g is not None:
                self.bank_g.finish()
            if lsc is not None:
                lsc.fetch()
        finally:
            for prm in self._d_params:
                prm.requires_grad_(True)
        self.optim_d.zero_grad()
        self._exchange(2, cut)
        if lsc is not None:
            lsc.decide(self.optim_g.opt_state)
            self.optim_g.flat_g.mul_(lsc.inv_scale)
        self.optim_g.step()
        out = {"loss_disc": loss_disc.detach(), "loss_gen": loss_gen.detach(), "loss_fm": loss_fm.detach(),
               "loss_mel": loss_mel.detach(), "kl_ssl": kl_ssl.detach(), "loss_kl": loss_kl.detach(),
               "grad_norm_d": self.optim_d.grad_norm(), "grad_norm_g": self.optim_g.grad_norm(),
               "loss_gen_all": loss_gen_all.detach()}
        if lsc is not None:
            # the scale this step ran with and the running totals of the fp16 range events (device scalars; see ops.DynamicLossScale)
            out.update({"loss_scale": lsc.scale.clone(), "f16_saturated": lsc.saturated, "f16_flushed": lsc.flushed,
                        "skipped_steps": lsc.skipped})
            lsc.update()
        return out


def build_parts(hps, device, seed=None, use_augment=None):
    """net_g, net_d, optim_g, optim_d, aug as `run` builds them (ttts/vqvae/train.py:175-205); rank-0 parameters are
    broadcast like DDP's constructor does (:207-208)."""
    torch.manual_seed(hps.train.seed if seed is None else seed)
    net_g = SynthesizerTrn(hps.data.filter_length // 2 + 1, hps.train.segment_size // hps.data.hop_length, **hps.vqvae).to(device)
    net_d = MultiPeriodDiscriminator(getattr(hps.vqvae, "use_spectral_norm", False)).to(device)
    if use_augment is None:
        use_augment = bool(getattr(hps.train, "augment", False))
    aug = Augment(hps).to(device) if use_augment else None                  # train.py:185
    tr = hps.train
    optim_g = FlatAdamW(net_g.parameters(), tr.learning_rate, betas=tr.betas, eps=tr.eps)
    optim_d = FlatAdamW(net_d.parameters(), tr.learning_rate, betas=tr.betas, eps=tr.eps)
    dp = FlatDataParallel()
    dp.broadcast_(optim_g.flat_p, optim_d.flat_p)
    if dp.enabled and dp.world > 1:   # dropout / sampling streams must differ across ranks (DDP leaves per-process RNG independent)
        from .attentions import _SeedSource
        _SeedSource.reseed(hps.train.seed if seed is None else seed, dp.rank)
        torch.manual_seed((hps.train.seed if seed is None else seed) + 7919 * dp.rank)
    net_g.train(); net_d.train()
    return net_g, net_d, optim_g, optim_d, aug, dp


class VqvaeTrainer:
    """Convenience owner of the parts + the step (tests, bench): `VqvaeTrainer(hps).train_step(batch)`."""

    def __init__(self, hps, device=None, seed=None, use_augment=None):
        self.rank, self.world, local = init_distributed()
        self.device = torch.device("cuda", local) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise ops.TttsError("ttts_amd.vqvae.train needs a GPU (no CPU fallback)")
        torch.cuda.set_device(self.device)
        self.hps = hps
        self.net_g, self.net_d, self.optim_g, self.optim_d, self.aug, self.dp = build_parts(hps, self.device, seed, use_augment)
        self.step_fn = VqvaeStep(hps, self.net_g, self.net_d, self.optim_g, self.optim_d, self.aug, self.dp)

    lr = property(lambda self: self.optim_g.lr,
                  lambda self, v: (setattr(self.optim_g, "lr", v), setattr(self.optim_d, "lr", v)) and None)

    def train_step(self, data, inject=None, cut=None, sync_buffers=True):
        ops.dropout_counter(self.device).add_(1)          # fresh dropout masks per step (also under graph replay)
        return self.step_fn(data, inject, cut=cut, sync_buffers=sync_buffers)

    # ---- the whole two-phase step as ONE hipGraph ----------------------------------------------------------------------------
    def train_step_graphed(self, data):
        """Replays the complete step (spectrograms, G forward, both D passes, six losses, both backward passes, both AdamW
        updates, codebook EMA: ~10 k kernel launches) from one captured hipGraph.  `data` is copied into static input
        buffers; the returned loss scalars are the graph's static outputs (read them before the next call).  Requirements:
        one fixed batch shape, codebook initialised.  With world size > 1 the step is recorded as three graphs around the two
        gradient all-reduces (see _capture).  The first call with a new
        shape runs two eager warm-up steps and records; if capture is refused the trainer says so ONCE and keeps running
        launch by launch."""
        # (the convolution precision is baked into a recording -- kernel selection, the loss-scale launches -- so it is part of the key)
        key = tuple((k, tuple(v.shape)) for k, v in sorted(data.items())) + (ops.conv_precision(),)
        st = getattr(self, "_graph_state", None)
        if st is None or st["key"] != key:
            # the old recording is dropped here: its frozen weight-gradient slab entries may follow the new batch shape again
            # (include/ttts_hip.h: ttts_conv_wgrad_arena_release_graphs) -- before the warm-up steps that re-shape them
            self._graph_state = None
            for a in self.step_fn.slabs_g + self.step_fn.slabs_d:
                a.release_graphs()
            st = self._capture(data, key)
            self._graph_state = st
        if st["graph"] is None:
            return self.train_step(data)
        for k, v in data.items():
            st["inputs"][k].copy_(v)
        if st.get("segments"):            # data-parallel: three recorded segments around the two gradient all-reduces
            self.step_fn._sync_buffers()
            for i, g in enumerate(st["segments"]):
                g.replay()
                if i < len(st["between"]):
                    st["between"][i]()
        else:
            st["graph"].replay()
        return st["out"]

    def _capture(self, data, key):
        cb = self.net_g.quantizer.vq.layers[0]._codebook
        if not bool(cb.inited):
            raise ops.TttsError("train_step_graphed: run the first (k-means initialising) step eagerly")
        from .quantize import EuclideanCodebook
        inputs = {k: v.clone() for k, v in data.items()}
        prev = EuclideanCodebook.sync_free
        EuclideanCodebook.sync_free = True                   # no host read-backs inside the step (see expire_codes_)
        try:
            # ONE warm-up stream per trainer, reused by every re-capture (a fresh torch.cuda.Stream per capture would make the
            # convolution contexts -- keyed by stream, 1.5 GB of operand scratch each -- grow with every new batch shape)
            side = getattr(self, "_warm_stream", None)
            if side is None:
                side = self._warm_stream = torch.cuda.Stream(device=self.device)
            warm_ok, warm_err = 1, None
            try:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                # warm-up on a side stream, as torch's capture recipe asks
                    for _ in range(2):
                        self.train_step(inputs)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
            except Exception as e:                           # noqa: BLE001
                warm_ok, warm_err = 0, e
            # every rank votes on every exit path: a rank that left here alone would skip the collectives its peers still issue
            if not (self.dp.all_ranks_ok(warm_ok) if self.dp.enabled else warm_ok):
                raise RuntimeError("warm-up failed on %s" % ("this rank: %s" % warm_err if warm_err is not None else "another rank"))
            if self.dp.enabled:
                # World size > 1: the collectives cannot be recorded, so the step is recorded as THREE graphs that share one
                # memory pool (tensors and the autograd graph of an earlier segment stay valid in the later ones) with the two
                # flat gradient all-reduces between them -- and the codebook broadcast in front -- issued eagerly at replay.
                # (Their overlap with compute is limited by the step itself: the generator phase's discriminator forward
                # needs the UPDATED discriminator, i.e. the first all-reduce, and nothing follows the second one.)
                # (thread-local capture mode: the process group's watchdog thread polls its work events while we record, which
                # the default global mode turns into hipErrorStreamCaptureUnsupported -- seen on the first RCCL run, world size 1)
                segs, between = [torch.cuda.CUDAGraph()], []
                ctx = [torch.cuda.graph(segs[0], capture_error_mode="thread_local")]

                def cut(fn):
                    ctx[0].__exit__(None, None, None)
                    between.append(fn)
                    fn()                                     # (keeps the ranks' collective sequences aligned during recording)
                    segs.append(torch.cuda.CUDAGraph())
                    ctx[0] = torch.cuda.graph(segs[-1], pool=segs[0].pool(), capture_error_mode="thread_local")
                    ctx[0].__enter__()
                self.step_fn._sync_buffers()
                ok, err = 1, None
                try:
                    ctx[0].__enter__()
                    try:
                        out = self.train_step(inputs, cut=cut, sync_buffers=False)
                    finally:
                        ctx[0].__exit__(None, None, None)
                except Exception as e:                       # noqa: BLE001 -- this rank's capture was refused
                    ok, err = 0, e
                    torch.cuda.synchronize()
                    arenas = [self.optim_d.flat_g, self.optim_g.flat_g]
                    while len(between) < 2:                  # keep the collective sequence aligned with the ranks that succeeded:
                        self.dp.allreduce_grads_(arenas[len(between)])   # the cuts not reached issue their all-reduce here (the
                        between.append(None)                 # sums are discarded: every rank then zeroes its gradients and runs eagerly)
                    self.optim_d.flat_g.zero_(); self.optim_g.flat_g.zero_()
                # the segments are only usable if EVERY rank recorded them: a rank that fell back to the eager step while its
                # peers replay three graphs would issue a different collective sequence
                if not self.dp.all_ranks_ok(ok):
                    raise RuntimeError("capture refused on %s" % ("this rank: %s" % err if err is not None else "another rank"))
                return {"key": key, "graph": segs[0], "segments": segs, "between": between, "inputs": inputs, "out": out}
            g = torch.cuda.CUDAGraph()
            alive = torch.distributed.is_available() and torch.distributed.is_initialized()
            with torch.cuda.graph(g, capture_error_mode="thread_local" if alive else "global"):
                out = self.train_step(inputs)
            return {"key": key, "graph": g, "inputs": inputs, "out": out}
        except Exception as err:                             # noqa: BLE001 -- any refusal: report once, run eagerly
            print("ttts_amd: hipGraph capture of the VQ-VAE-GAN step failed (%s); running it launch by launch"
                  % str(err).splitlines()[0][:200], file=sys.stderr, flush=True)
            EuclideanCodebook.sync_free = prev
            torch.cuda.synchronize()
            return {"key": key, "graph": None}

    def save(self, step):
        if self.rank != 0:
            return
        d = self.hps.train.exp_dir
        os.makedirs(d, exist_ok=True)
        save_checkpoint(self.net_g, self.optim_g, self.lr, step, os.path.join(d, "G_{}.pth".format(step)))
        save_checkpoint(self.net_d, self.optim_d, self.lr, step, os.path.join(d, "D_{}.pth".format(step)))

    def load_latest(self):
        d = self.hps.train.exp_dir
        _, _, _, it = load_checkpoint(latest_checkpoint_path(d, "D_*.pth"), self.net_d, self.optim_d)
        _, _, lr, it = load_checkpoint(latest_checkpoint_path(d, "G_*.pth"), self.net_g, self.optim_g)
        return it


class _NoScaler:
    """Stand-in for `GradScaler(enabled=False)` (train.py:297; fp16_run is False in the shipped config): the step body
    never scales, so the object only has to exist in the reference's argument list."""
    def scale(self, loss): return loss
    def unscale_(self, optimizer): pass
    def step(self, optimizer): optimizer.step()
    def update(self): pass


_steps = {}


def _step_for(hps, nets, optims, aug):
    key = (id(nets[0]), id(nets[1]), id(optims[0]), id(optims[1]))
    if key not in _steps:
        _steps.clear()
        _steps[key] = VqvaeStep(hps, nets[0], nets[1], optims[0], optims[1], aug)
    return _steps[key]


def train_and_evaluate(rank, epoch, hps, nets, optims, schedulers, scaler, loaders, logger, writers, aug):
    """One epoch, the reference's signature and flow (ttts/vqvae/train.py:298-470): `nets = [net_g, net_d]`,
    `optims = [optim_g, optim_d]` (FlatAdamW), `schedulers` / `scaler` untouched here as there (the caller steps the
    schedulers once per epoch; GradScaler is disabled), `loaders = [train_loader, eval_loader]` yielding VQVAECollater
    dicts, `writers = [writer, writer_eval]` or None, `aug` = Augment or None.  Losses are read back only every
    `log_interval` steps (the reference reads 12 + 1566 scalars per step)."""
    global global_step
    net_g, net_d = nets
    optim_g, optim_d = optims
    train_loader, eval_loader = loaders
    writer = writers[0] if writers is not None else None
    sampler = getattr(train_loader, "batch_sampler", None)
    if hasattr(sampler, "set_epoch"):
        sampler.set_epoch(epoch)
    step = _step_for(hps, nets, optims, aug)
    dev = next(net_g.parameters()).device
    net_g.train(); net_d.train()
    n_batches = len(train_loader) if hasattr(train_loader, "__len__") else None
    for batch_idx, data in enumerate(train_loader):
        data = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in data.items()}
        out = step(data)
        if rank == 0 and global_step % hps.train.log_interval == 0:
            lr = optim_g.param_groups[0]["lr"]
            vals = {k: float(v) for k, v in out.items()}
            head = "Train Epoch: {} [{:.0f}%]".format(epoch, 100.0 * batch_idx / n_batches if n_batches else 0.0)
            row = [vals[k] for k in ("loss_disc", "loss_gen", "loss_fm", "loss_mel", "kl_ssl", "loss_kl")] + [global_step, lr]
            if logger is not None:
                logger.info(head); logger.info(row)
            else:
                print(head, row, flush=True)
            if writer is not None:
                scalars = {"loss/g/total": vals["loss_gen_all"], "loss/d/total": vals["loss_disc"], "learning_rate": lr,
                           "grad_norm_d": vals["grad_norm_d"], "grad_norm_g": vals["grad_norm_g"], "loss/g/fm": vals["loss_fm"],
                           "loss/g/mel": vals["loss_mel"], "loss/g/kl_ssl": vals["kl_ssl"], "loss/g/kl": vals["loss_kl"]}
                for k, v in scalars.items():
                    writer.add_scalar(k, v, global_step)
        global_step += 1
    if epoch % hps.train.save_every_epoch == 0 and rank == 0:               # train.py:459-489
        tag = global_step if hps.train.if_save_latest == 0 else 233333333333
        os.makedirs(hps.train.exp_dir, exist_ok=True)
        save_checkpoint(net_g, optim_g, hps.train.learning_rate, epoch, os.path.join(hps.train.exp_dir, "G_{}.pth".format(tag)))
        save_checkpoint(net_d, optim_d, hps.train.learning_rate, epoch, os.path.join(hps.train.exp_dir, "D_{}.pth".format(tag)))
    if rank == 0 and logger is not None:
        logger.info("====> Epoch: {}".format(epoch))


class _EpochLoader:
    """`steps_per_epoch` batches of an endless source per pass, with the `batch_sampler.set_epoch` hook of the reference loader."""

    def __init__(self, source, steps_per_epoch):
        self.source, self.n = source, steps_per_epoch

    def __len__(self):
        return self.n

    def __iter__(self):
        for _ in range(self.n):
            yield next(self.source)


def run(rank, n_gpus, hps, train_loader=None, steps_per_epoch=100):
    """ttts/vqvae/train.py:119-296: build the parts, resume from the newest G_/D_ pair, ExponentialLR stepped once per
    epoch, `train_and_evaluate` per epoch.  One process per GPU (torchrun): `rank` / `n_gpus` are informational, the
    process group comes from the environment.  `train_loader`: any iterable of VQVAECollater dicts (default: the
    synthetic source, `dataset.path == "synthetic"`)."""
    global global_step
    r, world, local = init_distributed()
    device = torch.device("cuda", local)
    if not torch.cuda.is_available():
        raise ops.TttsError("ttts.vqvae.train needs a GPU (no CPU fallback)")
    torch.cuda.set_device(device)
    net_g, net_d, optim_g, optim_d, aug, dp = build_parts(hps, device)
    # everything built so far lives as long as the run: keep it out of the cyclic collector's passes (an eager step creates ~10^5
    # short-lived objects, and a full collection inside a step is a millisecond-scale gap in the launch stream)
    import gc
    gc.collect(); gc.freeze()
    if train_loader is None:
        if getattr(hps.dataset, "path", "synthetic") != "synthetic":
            raise NotImplementedError("ttts.vqvae.train ships the synthetic data source; pass train_loader= for real data "
                                      "(ttts_amd.vqvae.dataset.DistributedBucketSampler gives the reference's batching)")
        src = iter(SyntheticVqvaeBatches(hps.train.batch_size, n_samples=int(getattr(hps.dataset, "synthetic_samples", 163840)),
                                         text_len=int(getattr(hps.dataset, "synthetic_text_len", 64)), seed=hps.train.seed + r, device=device))
        train_loader = _EpochLoader(src, steps_per_epoch)
    try:
        _, _, _, epoch_str = load_checkpoint(latest_checkpoint_path(hps.train.exp_dir, "D_*.pth"), net_d, optim_d)
        _, _, _, epoch_str = load_checkpoint(latest_checkpoint_path(hps.train.exp_dir, "G_*.pth"), net_g, optim_g)
        global_step = (epoch_str - 1) * len(train_loader)
    except Exception:
        epoch_str, global_step = 1, 0
    for opt in (optim_g, optim_d):                                            # a resumed optimizer carries a decayed lr
        opt.param_groups[0]["lr"] = hps.train.learning_rate
        opt.param_groups[0].pop("initial_lr", None)
    scheduler_g = torch.optim.lr_scheduler.ExponentialLR(optim_g, gamma=hps.train.lr_decay, last_epoch=-1)
    scheduler_d = torch.optim.lr_scheduler.ExponentialLR(optim_d, gamma=hps.train.lr_decay, last_epoch=-1)
    for _ in range(epoch_str):
        scheduler_g.step(); scheduler_d.step()
    scaler = _NoScaler()
    for epoch in range(epoch_str, hps.train.epochs + 1):
        train_and_evaluate(r, epoch, hps, [net_g, net_d], [optim_g, optim_d], [scheduler_g, scheduler_d], scaler,
                           [train_loader, None], None, None, aug)
        scheduler_g.step(); scheduler_d.step()


def _spawned(local_rank, n_gpus, config_args, run_kwargs):
    """Body of one spawned rank (module-level so that the `spawn` start method can import it): the torchrun environment is
    written here, the process group itself comes up inside run() -> init_distributed()."""
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(n_gpus)})
    run(local_rank, n_gpus, get_hparams(*config_args), **run_kwargs)


def main(argv=None, **run_kwargs):
    """ttts/vqvae/train.py:44-60: one process per visible GPU, started from here.  Under a launcher (WORLD_SIZE in the
    environment: `torchrun --nproc-per-node N -m ttts.vqvae.train`) this process IS one rank and runs the body; otherwise --
    the reference's `python ttts/vqvae/train.py` -- `torch.cuda.device_count()` ranks are spawned on the loopback address and a
    free port (the reference draws a random port on "localhost"; the container's host name may not resolve), each running
    run(rank, n_gpus, hps).  One visible GPU: no process group, no spawn.  TTTS_SPAWN_RANKS overrides the rank count (tests:
    with TTTS_SHARE_GPU=1 the ranks share cuda:0 over gloo)."""
    argv = list(sys.argv[1:2] if argv is None else argv)
    if "WORLD_SIZE" in os.environ:
        return run(int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"]), get_hparams(*argv), **run_kwargs)
    if not torch.cuda.is_available():
        raise ops.TttsError("ttts.vqvae.train needs a GPU (no CPU fallback)")
    n_gpus = int(os.environ.get("TTTS_SPAWN_RANKS", "0")) or torch.cuda.device_count()
    if n_gpus <= 1:
        return run(0, 1, get_hparams(*argv), **run_kwargs)
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_spawned, nprocs=n_gpus, args=(n_gpus, argv, run_kwargs))


if __name__ == "__main__":
    main()
