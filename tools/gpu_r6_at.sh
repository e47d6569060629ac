#!/bin/bash
# bgemm with the XCD-contiguous tile order: parity (tests that use it), diffusion / VQ-VAE-GAN steps A/B vs HEAD's build, diffusion traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "attn or attention or bgemm or model or step or encoder or full" 2>&1 | tail -2
for rep in 1 2; do for lib in "" ttts_amd/libttts_hip_alt.so; do
  echo "lib=${lib:-in-tree}"; TTTS_LIB=$lib timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
  TTTS_LIB=$lib DFB_STEPS=30 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-60
done; done
bash tools/diffusion_pmc.sh 3 2>&1 | tail -13 | head -8 | cut -c1-160
