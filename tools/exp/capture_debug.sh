#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONFAULTHANDLER=1
for pools in "disc" "disc,synth" "disc,mrf" "disc,synth,mrf"; do
  TTTS_CAPTURE_POOLS=$pools timeout 200 python tools/exp/capture_debug.py ${1:-32} 2>&1 | grep "CAPTURE-OK\|Fatal\|Error" | head -3
done
