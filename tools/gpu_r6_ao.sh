#!/bin/bash
# what bounds the DMA kernel: build variants (tools/build_exp_lib.sh) timed per shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "" $(ls ttts_amd/libttts_hip_*.so | grep -v alt); do
  echo "== conv_bench B=32 lib=${lib:-in-tree}"
  TTTS_LIB=$lib CB_B=32 CB_ONLY="${CB_ONLY:-k5}" timeout 300 python tools/conv_bench.py 2>/dev/null | grep "GF" | cut -c1-100
done
