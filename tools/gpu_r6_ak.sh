#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_diffusion.py -q -p no:cacheprovider -x -k "conv1x1 or diffusion or wn_coupling or text_encoder" 2>&1 | tail -2
timeout 300 python tools/conv1x1_bench.py 2>/dev/null | tail -8 | cut -c1-120
for i in 1 2; do DFB_GRAPH=1 DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-60; done
for i in 1 2; do timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep CAPTURE-OK; done
