"""`python -m ttts_amd.gpt.train [config.json]` -- the reference's GPT trainer entry point
(ttts/gpt/train.py:41-145: `Trainer(cfg_path).train()`, `save(milestone)`, `load(model_path)`) on the HIP engine.

Same config keys (ttts/gpt/config.json), same loss weighting (:109), gradient accumulation (:99-112), clip 1.0 (:115),
AdamW / warm-up (:56-57), same checkpoint dict `{'step', 'model'}` (:70-77) and file naming `model-{step//1000}.pt`.
Differences, all on the host side of the hot loop:
  * one process per GPU under torchrun; gradients cross ranks as flat RCCL all-reduces of the gradient arena, the upper
    layers' half overlapped with the lower layers' backward (`parallel.FlatDataParallel`, `GptEngine.grad_exchange_plan`)
    instead of accelerate's bucketed DDP; the two `wait_for_everyone()` barriers per step (:117,121) are dropped --
    the all-reduce already orders the ranks;
  * losses / grad-norm are read back only every `val_freq` steps (the reference's `loss.item()` each micro-batch and
    84 `.item()` calls in get_grad_norm are per-step host stalls);
  * `dataset.path == "synthetic"` feeds seeded random batches of the collater's dict shape (gpt/dataset.py:91-97);
    real data loading (torchaudio / pypinyin / .vq.pth) is outside the hot path ("synthetic only");
  * tensorboard is optional (absent from the image): scalars also go to `train_log.jsonl`.
"""
import json
import os
import sys
import time
from datetime import datetime
from pathlib import Path

import torch

from ..parallel import FlatDataParallel, init_distributed
from .model import UnifiedVoice, prepare_tokens


def warmup(step):
    """ttts/gpt/train.py:36-40 (the engine applies the same rule on the device)."""
    return float(step / 500) if step < 500 else 1


def cycle(dl):
    """ttts/gpt/train.py:32-35."""
    while True:
        for data in dl:
            yield data


def get_grad_norm(model):
    """ttts/gpt/train.py:22-31 as ONE reduction over the flat gradient arena (the reference does 84 `.item()` reads);
    returns a Python float (one host sync)."""
    eng = model.engine
    state = torch.zeros(8, dtype=torch.float32, device=eng.device)
    from .. import ops
    ops.gradnorm(eng.grads, 0.0, state, eng.gn_ws)
    return float(state[4])


def _default_cfg_path():
    """The reference's default is the relative path 'ttts/gpt/config.json' (train.py:42); honour it when it exists in the
    working directory, otherwise use the packaged config."""
    return "ttts/gpt/config.json" if os.path.exists("ttts/gpt/config.json") else os.path.join(os.path.dirname(__file__), "config.json")


class SyntheticGptBatches:
    """Endless seeded batches with the keys of GptTtsCollater (gpt/dataset.py:91-97)."""

    def __init__(self, cfg, batch_size, text_len=128, mel_len=1024, seed=1234):
        self.c, self.B, self.tl, self.ml = cfg, batch_size, text_len, mel_len
        self.g = torch.Generator().manual_seed(seed)

    def __iter__(self):
        return self

    def __next__(self):
        comp = self.c.get("mel_length_compression", 1024)
        text = torch.randint(1, 255, (self.B, self.tl), generator=self.g, dtype=torch.int64)
        mel = torch.randint(0, self.c["start_mel_token"], (self.B, self.ml), generator=self.g, dtype=torch.int64)
        return {"padded_text": text, "text_lengths": torch.full((self.B,), self.tl, dtype=torch.int64),
                "padded_qmel": mel, "qmel_lengths": torch.full((self.B,), self.ml, dtype=torch.int64),
                "wav_lens": torch.full((self.B,), self.ml * comp, dtype=torch.int64)}


def clean_checkpoints(path_to_models, n_ckpts_to_keep=3):
    """Keep the newest n `model-*.pt` by mtime (ttts/utils/utils.py:67-85 with sort_by_time=True)."""
    ckpts = sorted(Path(path_to_models).glob("model-*.pt"), key=lambda f: f.stat().st_mtime)
    for f in ckpts[:max(0, len(ckpts) - n_ckpts_to_keep)]:
        f.unlink()


class Trainer(object):
    def __init__(self, cfg_path=None, dataloader=None, seed=0):
        cfg_path = _default_cfg_path() if cfg_path is None else cfg_path
        self.rank, self.world, self.local_rank = init_distributed()
        self.device = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.device)
        self.cfg = json.load(open(cfg_path))
        self.gpt = UnifiedVoice(**self.cfg["gpt"], device=self.device, seed=seed + self.rank)
        self.dp = FlatDataParallel()
        eng = self.gpt.engine
        self.dp.broadcast_(eng.params)          # accelerate.prepare -> DDP broadcasts rank-0 parameters
        eng.refresh_shadows()
        tr = self.cfg["train"]
        self.train_steps, self.val_freq = tr["train_steps"], tr["val_freq"]
        self.gradient_accumulate_every = tr["accumulate_num"]
        self.mel_loss_weight, self.text_loss_weight = tr["mel_weight"], tr["text_weight"]
        self.lr = tr["lr"]
        if dataloader is None:
            if self.cfg["dataset"]["path"] != "synthetic":
                raise NotImplementedError("ttts_amd.gpt.train ships the synthetic data source only; pass a dataloader "
                                          "yielding GptTtsCollater-style dicts for real data")
            dataloader = SyntheticGptBatches(self.gpt.cfg, self.cfg["dataloader"]["batch_size"], seed=1234 + self.rank)
        self.dataloader = iter(dataloader) if isinstance(dataloader, SyntheticGptBatches) else cycle(dataloader)   # train.py:59
        self.step = 0
        self.is_main = self.rank == 0
        if self.is_main:
            now = datetime.now()
            self.logs_folder = Path(tr["logs_folder"] + "/" + now.strftime("%Y-%m-%d-%H-%M-%S"))
            self.logs_folder.mkdir(exist_ok=True, parents=True)
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(log_dir=self.logs_folder)
            except Exception:
                self.writer = None

    # ---- checkpoints (dict layout of ttts/gpt/train.py:70-88) -------------------------------------------------
    def save(self, milestone):
        if not self.is_main:
            return
        eng = self.gpt.engine
        data = {"step": self.step, "model": {k: v.cpu() for k, v in eng.state_dict().items()},
                # beyond the reference's two keys (its loader ignores them): the AdamW moments, so that a resumed run
                # continues instead of restarting the optimizer
                "optimizer": {"exp_avg": eng.exp_avg.cpu(), "exp_avg_sq": eng.exp_avg_sq.cpu(), "step": float(eng.opt_state[0])}}
        torch.save(data, str(self.logs_folder / f"model-{milestone}.pt"))

    def load(self, model_path):
        """Reference checkpoints ({'step','model'}) resume like the reference does: weights + step counter, a FRESH AdamW
        and warm-up (train.py:56-57,79-88 never restore the optimizer).  Checkpoints written by `save` above also carry
        the moments and the optimizer step, and continue exactly."""
        data = torch.load(model_path, map_location="cpu")
        eng = self.gpt.engine
        self.step = data["step"]
        eng.load_state_dict(data["model"])
        opt = data.get("optimizer")
        if opt is not None and opt["exp_avg"].numel() == eng.exp_avg.numel():
            eng.exp_avg.copy_(opt["exp_avg"]); eng.exp_avg_sq.copy_(opt["exp_avg_sq"])
            eng.opt_state[0] = float(opt["step"])
        else:
            eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
            eng.opt_state[0] = 0.0
        eng.step_count = int(eng.opt_state[0])

    # ---- one optimizer step (ttts/gpt/train.py:96-121) -----------------------------------------------------------
    def train_step(self):
        eng = self.gpt.engine
        scale = self.dp.loss_scale() / self.gradient_accumulate_every
        opt = dict(lr=self.lr, max_norm=1.0, warmup_steps=500)
        if self.gradient_accumulate_every == 1:
            data = next(self.dataloader)
            if data is None:                      # the reference skips None batches but still steps the optimizer (:101-102,117-120)
                # same collective sequence as the peers that did get a batch (ranged exchange below): a rank that
                # issued one whole-arena all-reduce against their four ranged ones would hang or corrupt RCCL
                if self.dp.enabled:
                    _, first, second = eng.grad_exchange_plan()
                    for h in [self.dp.allreduce_range_(eng.grads, lo, hi) for lo, hi in first + second]:
                        if h is not None:
                            h.wait()
                eng.optimizer_step(**opt)
                eng.step_count += 1
                return
            toks = prepare_tokens(self.gpt.cfg, data["padded_text"], data["text_lengths"], data["padded_qmel"],
                                  data["wav_lens"])
            # data parallel: the gradients of the upper layers are all-reduced while the lower layers' backward runs
            # (GptEngine.grad_exchange_plan); launch by launch here because real batches change shape every step
            eng.train_step(toks, self.text_loss_weight * scale, self.mel_loss_weight * scale, capture=False,
                           exchange_range=(lambda lo, hi: self.dp.allreduce_range_(eng.grads, lo, hi)) if self.dp.enabled else None,
                           **opt)
            return
        for _ in range(self.gradient_accumulate_every):
            data = next(self.dataloader)
            if data is None:
                continue
            toks = prepare_tokens(self.gpt.cfg, data["padded_text"], data["text_lengths"], data["padded_qmel"],
                                  data["wav_lens"])
            eng.set_tokens(*toks)
            eng.forward()
            eng.backward(self.text_loss_weight * scale, self.mel_loss_weight * scale)
            eng.seed_ctr.add_(1)                 # every micro-batch draws its own dropout masks (forward + backward of one
                                                 # micro-batch share the value; optimizer_step advances it once more)
        self.dp.allreduce_grads_(eng.grads)      # accumulated micro-batches: ONE flat RCCL all-reduce (no-op at world size 1)
        eng.optimizer_step(**opt)
        eng.step_count += 1

    def train(self):
        eng = self.gpt.engine
        tr = self.cfg["train"]
        t0 = time.time()
        while self.step < self.train_steps:
            self.train_step()
            if self.is_main and self.step % self.val_freq == 0:
                lt, lm = eng.losses()
                st = eng.opt_state.tolist()
                rec = {"step": self.step, "loss": lt * self.text_loss_weight + lm * self.mel_loss_weight, "loss_mel": lm,
                       "loss_text": lt, "loss/grad": st[4], "lr": st[1], "elapsed_s": time.time() - t0}
                with open(self.logs_folder / "train_log.jsonl", "a") as f:
                    f.write(json.dumps(rec) + "\n")
                if self.writer is not None:
                    for k, v in rec.items():
                        self.writer.add_scalar(k, v, self.step)
                print(f"step {self.step} loss: {rec['loss']:.4f}", flush=True)
            if self.is_main and self.step % tr["save_freq"] == 0:
                if tr["keep_ckpts"] > 0:
                    clean_checkpoints(self.logs_folder, tr["keep_ckpts"])
                self.save(self.step // 1000)
            self.step += 1
        if self.is_main:
            print("training complete")


if __name__ == "__main__":
    trainer = Trainer(*sys.argv[1:2])
    trainer.train()
