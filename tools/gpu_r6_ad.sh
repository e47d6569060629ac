#!/bin/bash
# 64-row on-the-fly tiles: weight chunks prefetched a stage ahead (in-tree: K = 11 and K <= 3; alt_wpf: K = 5 / 7 too, at two waves) vs HEAD
cd "${GRAFT_REPO_ROOT:-/root/repo}"
touch ttts_amd/libttts_hip_alt_head.so ttts_amd/libttts_hip_alt_wpf.so     # (newer than the objects: lib.build() must not relink them)
for lib in "" ttts_amd/libttts_hip_alt_head.so ttts_amd/libttts_hip_alt_wpf.so; do
  echo "== conv_bench B=32 lib=${lib:-in-tree}"
  TTTS_LIB=$lib CB_B=32 CB_ONLY="RB1" timeout 300 python tools/conv_bench.py 2>/dev/null | grep "RB1(128)\|RB1(256)\|RB1(64)" | cut -c1-112
done
for rep in 1 2; do for lib in "" ttts_amd/libttts_hip_alt_head.so ttts_amd/libttts_hip_alt_wpf.so; do
  echo "graph step lib=${lib:-in-tree}"; TTTS_LIB=$lib timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
done; done
timeout 900 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "conv or generator or resblock or posterior or full_vqvae" 2>&1 | tail -2
