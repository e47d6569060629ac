#!/bin/bash
# XCD-aware tile order in the 1 x 1 / on-the-fly convolution kernels and the weight-gradient kernels: parity, steps A/B vs HEAD's build, fabric traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_diffusion.py -q -p no:cacheprovider -x 2>&1 | tail -2
for rep in 1 2; do for lib in "" ttts_amd/libttts_hip_alt.so; do
  echo "lib=${lib:-in-tree}"; TTTS_LIB=$lib timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
  TTTS_LIB=$lib DFB_STEPS=30 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-200
done; done
bash tools/diffusion_pmc.sh 3 2>&1 | tail -14 | cut -c1-160
