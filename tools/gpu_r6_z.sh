#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for lib in "" ttts_amd/libttts_hip_alt.so; do
  echo "== conv_bench B=32 lib=${lib:-in-tree}"
  TTTS_LIB=$lib CB_B=32 CB_ONLY="k11" timeout 300 python tools/conv_bench.py 2>/dev/null | grep "RB1" | cut -c100-150
done; done
