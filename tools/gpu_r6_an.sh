#!/bin/bash
# K = 5 at 64-row tiles: first weight chunks requested above the stage barrier -- parity, per-shape A/B against HEAD's build, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "conv or disc or wn or full_vqvae" 2>&1 | tail -2
for lib in "" ttts_amd/libttts_hip_alt.so; do
  echo "== conv_bench B=32 lib=${lib:-in-tree}"
  TTTS_LIB=$lib CB_B=32 CB_ONLY="k5" timeout 300 python tools/conv_bench.py 2>/dev/null | grep "k5" | cut -c1-130
done
for rep in 1 2; do for lib in "" ttts_amd/libttts_hip_alt.so; do
  echo "graph step lib=${lib:-in-tree}"; TTTS_LIB=$lib timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
done; done
