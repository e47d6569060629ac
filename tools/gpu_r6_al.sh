#!/bin/bash
# fused-taps weight gradient with the masks at deposit time (no per-pair waits): parity + A/B against HEAD
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "conv or generator or resblock or posterior or full_vqvae" 2>&1 | tail -2
for lib in "" ttts_amd/libttts_hip_alt.so; do
  echo "== conv_bench B=32 lib=${lib:-in-tree}"
  TTTS_LIB=$lib CB_B=32 CB_ONLY="RB1" timeout 300 python tools/conv_bench.py 2>/dev/null | grep "RB1" | cut -c1-32,108-150
done
for rep in 1 2; do for lib in "" ttts_amd/libttts_hip_alt.so; do
  echo "graph step lib=${lib:-in-tree}"; TTTS_LIB=$lib timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
done; done
