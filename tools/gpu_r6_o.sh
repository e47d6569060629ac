#!/bin/bash
# recorded step: ResBlock pairs as one autograd node (no engine-side gradient adds); eager: does the ResBlock fan-out still pay?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONFAULTHANDLER=1
for rp in 0 1 0 1; do echo "graph TTTS_RESPAIR=$rp"; TTTS_RESPAIR=$rp timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3; done
for ep in "disc,synth,mrf" "disc,synth"; do for rp in 0 1; do
  echo "eager TTTS_EAGER_POOLS=$ep TTTS_RESPAIR=$rp"; TTTS_RESPAIR=$rp TTTS_EAGER_POOLS=$ep timeout 300 python tools/vqvae_bench.py 32 10 3 2>/dev/null | tail -1 | cut -c1-160
done; done
