"""ttts_amd -- MI355X-native (gfx950) training hot path of adelacvg/ttts.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed = RCCL); every hot operator is a
hand-written HIP kernel in `libttts_hip.so`, reached through the C ABI declared in `include/ttts_hip.h`
(`ttts_amd.lib`).  There is no CPU / eager fallback: using an operator without the built library raises.
"""
__version__ = "0.1.0"
