#!/bin/bash
# XCD-aware order: which kernels pay in the VQ-VAE-GAN step -- in-tree (all) vs without the on-the-fly kernel's remap vs HEAD; conv-family traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do for lib in "" ttts_amd/libttts_hip_nootf.so ttts_amd/libttts_hip_alt.so; do
  echo "lib=${lib:-in-tree}"; TTTS_LIB=$lib timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
done; done
TTTS_BRANCH_STREAMS=0 TTTS_D_STREAMS=0 bash tools/vqvae_pmc.sh 1 2>&1 | tail -14 | cut -c1-160
