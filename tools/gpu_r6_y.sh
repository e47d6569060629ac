#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_fullsize.py tests/test_gpu_diffusion.py -q -p no:cacheprovider -x 2>&1 | tail -2
for lib in "" ttts_amd/libttts_hip_alt.so; do
  echo "== conv_bench B=32 lib=${lib:-in-tree}"
  TTTS_LIB=$lib CB_B=32 CB_ONLY="RB1" timeout 300 python tools/conv_bench.py 2>/dev/null | grep "RB1" | cut -c1-150
done
for rep in 1 2; do for lib in "" ttts_amd/libttts_hip_alt.so; do
  echo "graph step lib=${lib:-in-tree}"; TTTS_LIB=$lib timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
done; done
for g in 0 1; do DFB_GRAPH=$g DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-120; done
