#!/bin/bash
# round 6, second call: world-1 RCCL tests after the capture-mode fix, the ABI v11 tests, then the whole vqvae / fp8 / diffusion files
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gpt.py -q -p no:cacheprovider -x -k "world1 or rccl" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "nan or dynamic_loss or refused or tf32class" 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_fp8.py tests/test_gpu_diffusion.py tests/test_gpu_kernels.py -q -p no:cacheprovider -x 2>&1 | tail -8
