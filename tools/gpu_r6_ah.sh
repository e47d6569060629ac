#!/bin/bash
# fused 1 x 1 weight gradient: parity + A/B (TTTS_DEBUG_FLAGS=2048 = the pre-pass + pre-split kernel path)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_diffusion.py tests/test_gpu_fp8.py -q -p no:cacheprovider -x 2>&1 | tail -2
for fl in 0 2048; do echo "== conv_bench flags=$fl"; CB_B=32 CB_ONLY="k1" timeout 300 python tools/conv_bench.py $fl 2>/dev/null | grep "k1" | cut -c1-150; done
for rep in 1 2; do for fl in 0 2048; do
  echo "graph step TTTS_DEBUG_FLAGS=$fl"; TTTS_DEBUG_FLAGS=$fl timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
  echo "diffusion TTTS_DEBUG_FLAGS=$fl"; TTTS_DEBUG_FLAGS=$fl DFB_GRAPH=1 DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-60
done; done
