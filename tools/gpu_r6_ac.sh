#!/bin/bash
# dispatch threshold of the pre-split + DMA path: from 3 output-channel tiles (default) vs from 1 (flag 32768), per ResBlock shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fl in 0 32768; do echo "== flags=$fl"; CB_B=32 CB_ONLY="RB1" timeout 300 python tools/conv_bench.py $fl 2>/dev/null | grep "RB1" | cut -c1-112; done
