#!/bin/bash
# 192-channel WaveNet k5 layers on the DMA kernel: tile choice (262144 = always 64 x 256), stages (268435456 = one block per stage)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fl in 0 262144 268435456 0 262144; do
  echo "== conv_bench B=32 flags=$fl"
  CB_B=32 CB_ONLY="WN in" timeout 300 python tools/conv_bench.py $fl 2>/dev/null | grep "GF" | cut -c1-130
done
