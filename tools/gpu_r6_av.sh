#!/bin/bash
# 1024 -> 1024 k5 at 528 workgroups of 64 x 256 (one round + 16): the 64 x 128 tile (flag 131072) for comparison
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fl in 0 131072 0 131072; do
  echo "== conv_bench B=32 flags=$fl"
  CB_B=32 CB_ONLY="1024->1024" timeout 300 python tools/conv_bench.py $fl 2>/dev/null | grep "GF" | cut -c1-100
done
