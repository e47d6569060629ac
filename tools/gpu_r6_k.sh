#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for gmode in 0 1; do echo "fp8 DFB_GRAPH=$gmode"; TTTS_DIFFUSION_PRECISION=fp8 DFB_GRAPH=$gmode DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-250; done
echo "fp8+tf32class graph"; TTTS_CONV_PRECISION=tf32class TTTS_DIFFUSION_PRECISION=fp8 DFB_GRAPH=1 DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_fp8.py -q -p no:cacheprovider -x -k "tf32class or dynamic_loss or nan or fp8" 2>&1 | grep -v "^$" | tail -3
timeout 300 env TTTS_CONV_PRECISION=tf32class python tools/vqvae_bench.py 32 8 2 2>/dev/null | tail -1 | cut -c1-200
