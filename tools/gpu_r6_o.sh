#!/bin/bash
# recorded / eager step: ResBlock pairs as one autograd node, now that the gate + residual epilogue is batched
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rp in 0 1 0 1; do echo "graph TTTS_RESPAIR=$rp"; TTTS_RESPAIR=$rp timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3; done
for rp in 0 1; do echo "eager TTTS_RESPAIR=$rp"; TTTS_RESPAIR=$rp timeout 300 python tools/vqvae_bench.py 32 10 3 2>/dev/null | tail -1 | cut -c1-160; done
TTTS_RESPAIR=1 timeout 900 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "generator or resblock or posterior or full_vqvae" 2>&1 | tail -2
