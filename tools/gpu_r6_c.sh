#!/bin/bash
# round 6, third call: the fused relative-position attention (tests, then the diffusion leg A/B), plus the fixed world-1 / loss-scale tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fp8.py -q -p no:cacheprovider -x 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_gpt.py -q -p no:cacheprovider -x -k "world1 or rccl" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "dynamic_loss" 2>&1 | tail -5
for f in 1 0; do
  echo "TTTS_DIFFUSION_FUSED_ATTN=$f"
  TTTS_DIFFUSION_FUSED_ATTN=$f timeout 300 python tools/diffusion_bench.py 2>&1 | tail -3
done
