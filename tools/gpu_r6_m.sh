#!/bin/bash
# sibling fan-outs: (1) the recorded step with every level, (2) eager step with the posterior branch inline vs on a side stream,
# (3) census of the ATen kernels still on the VQ-VAE-GAN / diffusion product path
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONFAULTHANDLER=1
O=gpurun_out/r6m; mkdir -p $O
timeout 300 python tools/exp/capture_debug.py 32 > $O/cap_all.txt 2>&1; echo "capture all levels rc=$?"; grep "CAPTURE-OK\|Fatal\|Error\|failed" $O/cap_all.txt | head -5
TTTS_CAPTURE_POOLS=disc,synth timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
for inl in 1 -1 1 -1; do
  echo "eager TTTS_SYNTH_INLINE=$inl"; TTTS_SYNTH_INLINE=$inl timeout 300 python tools/vqvae_bench.py 32 10 3 2>/dev/null | tail -1 | cut -c1-160
done
timeout 300 python tools/exp/aten_census.py 32 vqvae > $O/aten_vqvae.txt 2>&1; echo "census rc=$?"; head -3 $O/aten_vqvae.txt
timeout 300 python tools/exp/aten_census.py 16 diffusion > $O/aten_diffusion.txt 2>&1; echo "census rc=$?"; head -3 $O/aten_diffusion.txt
