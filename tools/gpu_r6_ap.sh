#!/bin/bash
# phase-merged strided data gradients with short rows as one virtual row: parity, per-shape A/B (flag 524288 = segment folding), step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "conv or disc or strided or full_vqvae" 2>&1 | tail -3
for fl in 0 524288 0 524288; do
  echo "== conv_bench B=32 flags=$fl"
  CB_B=32 CB_ONLY="s3" timeout 300 python tools/conv_bench.py $fl 2>/dev/null | grep "GF" | cut -c1-130
done
for rep in 1 2; do for fl in 0 524288; do
  echo "graph step flags=$fl"; TTTS_DEBUG_FLAGS=$fl timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
done; done
