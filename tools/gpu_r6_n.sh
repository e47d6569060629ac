#!/bin/bash
# recorded VQ-VAE-GAN step: which fan-out levels pay under replay?  + ATen census of the eager step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONFAULTHANDLER=1
O=gpurun_out/r6n; mkdir -p $O
for pools in "none" "disc" "synth" "disc,synth" "disc,mrf" "synth,mrf" "disc,synth,mrf" "disc,synth"; do
  TTTS_CAPTURE_POOLS=$pools timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
done
for ds in 2 6; do echo "TTTS_D_STREAMS=$ds"; TTTS_D_STREAMS=$ds TTTS_CAPTURE_POOLS=disc,synth timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3; done
timeout 300 python tools/exp/aten_census.py 32 vqvae > $O/aten_vqvae.txt 2>&1; echo "census rc=$?"; grep -A3 "^ATEN" $O/aten_vqvae.txt
