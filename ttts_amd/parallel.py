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


def dp_forced():
    """TTTS_DP_FORCE=1: run the data-parallel exchange (process group, collectives, the graphs captured beside the live
    communicator) even at world size 1.  A sum over one rank is the identity, so results are bit-identical to the
    non-distributed step -- what the flag buys is that RCCL (backend "nccl") really executes on a 1-GPU box."""
    return os.environ.get("TTTS_DP_FORCE", "0") not in ("", "0")


def init_distributed(backend=None):
    """Initialise the default process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    rank, world, local = env_rank_world()
    if os.environ.get("TTTS_SHARE_GPU"):   # test hook: several ranks on ONE GPU (RCCL refuses duplicate GPUs -> gloo)
        local = 0
        backend = backend or "gloo"
    if (world > 1 or dp_forced()) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("TTTS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world == 1:                      # forced single rank with no launcher: any free loopback port
                import socket
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            else:
                os.environ["MASTER_PORT"] = "29500"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class _WidenOnWait:
    """Handle of a reduced-precision ranged all-reduce: .wait() completes the collective and widens the sum back into the fp32 range."""

    def __init__(self, work, staged, dst):
        self.work, self.staged, self.dst = work, staged, dst

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.staged)


class FlatDataParallel:
    """Gradient / parameter exchange for a replica whose parameters and gradients are single flat tensors."""

    def __init__(self, group=None, grad_dtype=None):
        """grad_dtype: torch.bfloat16 sends the gradient exchange in bf16 (half the bytes on the xGMI links: 43 instead of 86 MB
        for the GPT arena); the fp32 arena is rounded into a staging buffer, summed there and widened back, so every rank ends
        with the SAME fp32 values (replicas stay bit-identical), each within bf16 rounding (2^-8 relative per addend and partial sum) of the
        fp32 sum.  Default (None, or TTTS_DP_GRAD_DTYPE unset): fp32, the reference DDP's arithmetic."""
        self.group = group
        if grad_dtype is None and os.environ.get("TTTS_DP_GRAD_DTYPE", "") in ("bf16", "bfloat16"):
            grad_dtype = torch.bfloat16
        self.grad_dtype = grad_dtype
        self._stages = {}                    # one bf16 staging buffer per arena (keyed by its data pointer): D and G alternate
        self.enabled = (dist.is_available() and dist.is_initialized()
                        and (dist.get_world_size(group) > 1 or dp_forced()))
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
        if self.grad_dtype is not None and flat_grads.dtype != self.grad_dtype:
            h = self.allreduce_range_(flat_grads, 0, flat_grads.numel())
            if async_op:
                return h
            h.wait()
            return None
        return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def allreduce_range_(self, flat_grads, lo, hi):
        """Asynchronous SUM all-reduce of flat_grads[lo:hi]; returns a handle with .wait() (None when not distributed)."""
        if not self.enabled or hi <= lo:
            return None
        if self.grad_dtype is not None and flat_grads.dtype != self.grad_dtype:
            key = (flat_grads.data_ptr(), flat_grads.numel(), str(flat_grads.device))
            stage = self._stages.get(key)
            if stage is None:
                stage = self._stages[key] = torch.empty(flat_grads.numel(), dtype=self.grad_dtype, device=flat_grads.device)
            st = stage[lo:hi]
            st.copy_(flat_grads[lo:hi])                      # (on the collective's input stream order: torch.distributed syncs with it)
            work = dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return _WidenOnWait(work, st, flat_grads[lo:hi])
        return dist.all_reduce(flat_grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_ranks_ok(self, ok):
        """True iff `ok` is truthy on EVERY rank (a MIN all-reduce of a flag).  Used where a rank-local failure must become a
        collective decision -- e.g. hipGraph capture refused on one rank: its peers must not replay graphs whose collective
        sequence the failed rank will not issue."""
        if not self.enabled:
            return bool(ok)
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

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
