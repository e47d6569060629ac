"""Stand-alone timing of the fused residual-GEMM + LayerNorm launch against the two launches it replaces (hipGraph-replayed, HIP events)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from ttts_amd import ops
from ttts_amd.lib import EPI_RESID_ADD_F32
dev = torch.device("cuda:0")
M, N = 9248, 512
g = torch.Generator().manual_seed(0)
for K in (512, 2048):
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev); w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev); resid = torch.randn(M, N, generator=g).to(dev)
    gamma = torch.ones(N, device=dev); beta = torch.zeros(N, device=dev)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    x = torch.empty(M, N, device=dev); y = torch.empty(M, N, dtype=torch.bfloat16, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)

    def two():
        ops.gemm_nt(a, w, x, bias, epilogue=EPI_RESID_ADD_F32, resid_in=resid, dropout_p=0.1, seed=3, counter=ctr)
        ops.layernorm_fwd(x, gamma, beta, y, mean, rstd)

    def one():
        ops.gemm_nt_resid_ln(a, w, x, gamma, beta, y, mean, rstd, bias=bias, resid_in=resid, dropout_p=0.1, seed=3, counter=ctr)
    print("K = %d: two launches %.1f us, fused %.1f us" % (K, bench._event_time_us(two, 20), bench._event_time_us(one, 20)))
