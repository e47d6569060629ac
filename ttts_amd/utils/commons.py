"""The helpers of ttts/utils/commons.py that the training path calls (`slice_segments` :48-54, `rand_slice_segments`
:57-66, `sequence_mask` :125-130, `clip_grad_value_` :148-163), on the HIP path."""
import torch

from ..vqvae.vq2 import rand_slice_segments, sequence_mask, slice_segments  # noqa: F401


def clip_grad_value_(parameters, clip_value, norm_type=2):
    """Total gradient 2-norm of `parameters` (commons.py:148-163; the reference's trainer calls it with clip_value=None,
    i.e. as a norm only).  The parameters must be the ones a `FlatAdamW` re-homed: the norm is one reduction over the flat
    gradient arena instead of one `.item()` per tensor.  Returns a Python float like the reference (one host sync)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = [p for p in parameters if p.grad is not None]
    if clip_value is not None or float(norm_type) != 2.0:
        raise NotImplementedError("clip_grad_value_: the path only measures the 2-norm (clip_value=None)")
    from .. import ops
    bases = {}
    for p in parameters:
        base = p.grad._base if p.grad._base is not None else p.grad
        bases[base.data_ptr()] = base
    total = 0.0
    for base in bases.values():
        flat = base.reshape(-1)
        state = torch.zeros(8, dtype=torch.float32, device=flat.device)
        ops.gradnorm(flat, 0.0, state, ops.gradnorm_workspace(flat.numel(), flat.device))
        total += float(state[4]) ** 2
    return total ** 0.5
