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
from .augment import Augment, augment
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


class VqvaeTrainer:
    """Owns net_g / net_d / the two optimizers and runs the two-phase step."""

    def __init__(self, hps, device=None, seed=None, use_augment=None):
        self.rank, self.world, local = init_distributed()
        self.device = torch.device("cuda", local) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise ops.TttsError("ttts_amd.vqvae.train needs a GPU (no CPU fallback)")
        torch.cuda.set_device(self.device)
        self.hps = hps
        torch.manual_seed(hps.train.seed if seed is None else seed)
        self.net_g = SynthesizerTrn(hps.data.filter_length // 2 + 1, hps.train.segment_size // hps.data.hop_length,
                                    **hps.vqvae).to(self.device)
        self.net_d = MultiPeriodDiscriminator(getattr(hps.vqvae, "use_spectral_norm", False)).to(self.device)
        self.dp = FlatDataParallel()
        if use_augment is None:
            use_augment = bool(getattr(hps.train, "augment", False))
        self.aug = Augment(hps).to(self.device) if use_augment else None      # train.py:185
        tr = hps.train
        self.optim_g = FlatAdamW(self.net_g.parameters(), tr.learning_rate, betas=tr.betas, eps=tr.eps)
        self.optim_d = FlatAdamW(self.net_d.parameters(), tr.learning_rate, betas=tr.betas, eps=tr.eps)
        self.dp.broadcast_(self.optim_g.flat_p, self.optim_d.flat_p)        # DDP construction: rank-0 parameters
        self.lr = tr.learning_rate
        self.net_g.train(); self.net_d.train()

    def _sync_buffers(self):
        if self.dp.enabled:                                                  # DDP broadcast_buffers=True (rank-0 codebook)
            self.dp.broadcast_(*[b for b in self.net_g.buffers() if b.is_floating_point()])

    def train_step(self, data, inject=None):
        """data: dict(wav (B, T) f32, wav_lengths, text, text_lengths) on the device.  Returns a dict of device scalars."""
        h, tr = self.hps.data, self.hps.train
        inject = dict(inject or {})
        wav, wav_lengths, text, text_lengths = data["wav"], data["wav_lengths"], data["text"], data["text_lengths"]
        y = wav
        spec = spectrogram_torch(wav, h.filter_length, h.hop_length, h.win_length, center=False)
        spec_lengths = torch.div(wav_lengths, h.hop_length, rounding_mode="floor")
        if "wav_aug" in inject:
            wav_aug = inject.pop("wav_aug")                                   # tests: a fixed augmented clip
        elif self.aug is None:
            wav_aug = wav                                                     # train.py:335-336
        else:
            wav_aug = augment(wav, self.aug, self.hps)                        # train.py:337-338 (PEQ part)
        spec_aug = spec if wav_aug is wav else spectrogram_torch(wav_aug, h.filter_length, h.hop_length, h.win_length, center=False)
        self._sync_buffers()
        y_hat, kl_ssl, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), quantized = self.net_g(
            wav, wav_aug, wav_lengths, spec, spec_aug, spec_lengths, text, text_lengths, **inject)
        mel = spec_to_mel_torch(spec, h.filter_length, h.n_mel_channels, h.sampling_rate, h.mel_fmin, h.mel_fmax)
        y_mel = slice_segments(mel, ids_slice, tr.segment_size // h.hop_length)
        y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1), h.filter_length, h.n_mel_channels, h.sampling_rate, h.hop_length,
                                          h.win_length, h.mel_fmin, h.mel_fmax)
        y = slice_segments(y.unsqueeze(1), ids_slice * h.hop_length, tr.segment_size)
        scale = self.dp.loss_scale()
        # ---- discriminator phase
        y_d_hat_r, y_d_hat_g, _, _ = self.net_d(y, y_hat.detach())
        loss_disc, losses_disc_r, losses_disc_g = L.discriminator_loss(y_d_hat_r, y_d_hat_g)
        self.optim_d.zero_grad()
        (loss_disc * scale).backward()
        self.dp.allreduce_grads_(self.optim_d.flat_g)
        self.optim_d.step(self.lr)
        # ---- generator phase
        y_d_hat_r, y_d_hat_g, fmap_r, fmap_g = self.net_d(y, y_hat)
        loss_mel = L.l1_loss(y_mel, y_hat_mel) * tr.c_mel
        loss_kl = L.kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * tr.c_kl
        loss_fm = L.feature_loss(fmap_r, fmap_g)
        loss_gen, losses_gen = L.generator_loss(y_d_hat_g)
        loss_gen_all = loss_gen + loss_fm + loss_mel + kl_ssl * 1 + loss_kl
        self.optim_g.zero_grad()
        (loss_gen_all * scale).backward()
        self.optim_d.zero_grad()                                             # the G backward also reached net_d's arena
        self.dp.allreduce_grads_(self.optim_g.flat_g)
        self.optim_g.step(self.lr)
        return {"loss_disc": loss_disc.detach(), "loss_gen": loss_gen.detach(), "loss_fm": loss_fm.detach(),
                "loss_mel": loss_mel.detach(), "kl_ssl": kl_ssl.detach(), "loss_kl": loss_kl.detach(),
                "grad_norm_d": self.optim_d.grad_norm(), "grad_norm_g": self.optim_g.grad_norm(),
                "loss_gen_all": loss_gen_all.detach()}

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
        _, _, self.lr, it = load_checkpoint(latest_checkpoint_path(d, "G_*.pth"), self.net_g, self.optim_g)
        return it


def train_and_evaluate(rank, epoch, hps, trainer, loader, steps_per_epoch, logger=None):
    """One epoch of the step body (train.py:298-406); `loader` yields collater dicts."""
    global global_step
    for batch_idx in range(steps_per_epoch):
        data = next(loader)
        out = trainer.train_step(data)
        if rank == 0 and global_step % hps.train.log_interval == 0:
            vals = {k: float(v) for k, v in out.items()}
            msg = "Train Epoch: {} [{:.0f}%] step {} lr {:.3e} {}".format(epoch, 100.0 * batch_idx / steps_per_epoch,
                                                                          global_step, trainer.lr, json.dumps(vals))
            (logger.info if logger else print)(msg)
        if rank == 0 and global_step % hps.train.save_freq == 0 and global_step > 0:
            trainer.save(global_step)
        global_step += 1


def run(rank, n_gpus, hps, steps_per_epoch=100):
    global global_step
    trainer = VqvaeTrainer(hps)
    try:
        it = trainer.load_latest()
        global_step = it
        epoch_str = it // steps_per_epoch + 1
    except Exception:
        epoch_str, global_step = 1, 0
    loader = iter(SyntheticVqvaeBatches(hps.train.batch_size, seed=hps.train.seed + trainer.rank, device=trainer.device))
    for epoch in range(epoch_str, hps.train.epochs + 1):
        trainer.lr = hps.train.learning_rate * hps.train.lr_decay ** epoch    # ExponentialLR stepped once per epoch
        train_and_evaluate(trainer.rank, epoch, hps, trainer, loader, steps_per_epoch)


def main():
    hps = get_hparams(*sys.argv[1:2])
    if getattr(hps.dataset, "path", "synthetic") != "synthetic":
        raise NotImplementedError("ttts_amd.vqvae.train ships the synthetic data source only")
    n_gpus = int(os.environ.get("WORLD_SIZE", "1"))
    run(int(os.environ.get("RANK", "0")), n_gpus, hps)


if __name__ == "__main__":
    main()
