"""Data-parallel plumbing: one process per GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI on ROCm)
on the GPU box and "gloo" in the CPU tests.  Replaces accelerate's DDP wrapper (ttts/gpt/train.py:43,58,112,117,121)
and torch DDP (ttts/vqvae/train.py:127-132,207-208).

The path shards by data only: every rank holds a full replica and its own micro-batch; the single
exchange step is a SUM all-reduce of the flat fp32 gradient arena (the 1/world factor is folded into the loss
weights, so no extra scaling pass).  One large collective instead of 25 MB DDP buckets: xGMI is point-to-point
(7 links x ~153 GB/s per GPU), RCCL picks a direct algorithm for a message of this size, and the GPT backward is
only a few ms long, so there is little to overlap with.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None):
    """Initialise the default process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    rank, world, local = env_rank_world()
    if os.environ.get("TTTS_SHARE_GPU"):   # test hook: several ranks on ONE GPU (RCCL refuses duplicate GPUs -> gloo)
        local = 0
        backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("TTTS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class FlatDataParallel:
    """Gradient / parameter exchange for a replica whose parameters and gradients are single flat tensors."""

    def __init__(self, group=None):
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0

    def loss_scale(self):
        """Multiply the loss weights by this so that a SUM all-reduce yields the data-parallel MEAN gradient."""
        return 1.0 / self.world

    def broadcast_(self, *tensors, src=0):
        """Make every replica start from rank-`src` state (parameters, Adam moments, codebook buffers)."""
        if self.enabled:
            for t in tensors:
                dist.broadcast(t, src=src, group=self.group)

    def allreduce_grads_(self, flat_grads, async_op=False):
        if not self.enabled:
            return None
        return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def allreduce_range_(self, flat_grads, lo, hi):
        """Asynchronous SUM all-reduce of flat_grads[lo:hi]; returns a handle with .wait() (None when not distributed)."""
        if not self.enabled or hi <= lo:
            return None
        return dist.all_reduce(flat_grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def barrier(self):
        if self.enabled:
            dist.barrier(group=self.group)

    def max_over_ranks(self, value):
        if not self.enabled:
            return value
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())


def shard_indices(n_items, rank, world):
    """Rank-strided sharding `ids[rank::world]`, the reference sampler's rule (ttts/vqvae/dataset.py:277)."""
    return list(range(n_items))[rank::world]
