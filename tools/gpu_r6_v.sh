#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x 2>&1 | tail -3
bash tools/gpu_r6_u.sh
for rep in 1 2; do timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3; done
timeout 300 python tools/vqvae_bench.py 32 10 3 2>/dev/null | tail -1 | cut -c1-160
CB_B=32 timeout 300 python tools/conv_bench.py 2>/dev/null | grep "RB1\|WN\|Disc\|FFN"
