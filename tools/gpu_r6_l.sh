#!/bin/bash
# nested fan-out under capture: which pool combinations crash hipStreamEndCapture with the round-6 code? (one pool set per process)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONFAULTHANDLER=1
mkdir -p gpurun_out/r6l
for pools in ${POOLSETS:-"mrf" "disc,mrf" "synth,mrf"}; do
  TTTS_CAPTURE_POOLS=$pools timeout 300 python tools/exp/capture_debug.py 32 > gpurun_out/r6l/cap_${pools//,/_}.txt 2>&1
  echo "pools=$pools rc=$?"; grep "CAPTURE-OK\|Fatal\|Error\|failed" gpurun_out/r6l/cap_${pools//,/_}.txt | head -5
done
