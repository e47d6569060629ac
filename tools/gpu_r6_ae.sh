#!/bin/bash
# conv1x1_b3_kernel with 128-row tiles for wide layers: parity, per-shape and diffusion-step A/B (TTTS_DEBUG_FLAGS=1024 = off)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_diffusion.py -q -p no:cacheprovider -x -k "conv1x1 or diffusion" 2>&1 | tail -2
for fl in 0 1024; do echo "== conv1x1_bench TTTS_DEBUG_FLAGS=$fl"; TTTS_DEBUG_FLAGS=$fl timeout 300 python tools/conv1x1_bench.py 2>/dev/null | tail -8; done
for rep in 1 2; do for fl in 0 1024; do echo "diffusion TTTS_DEBUG_FLAGS=$fl"; TTTS_DEBUG_FLAGS=$fl DFB_GRAPH=1 DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-100; done; done
