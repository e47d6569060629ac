#!/bin/bash
# round 4: WeightNormBank -- parity tests, then a same-box A/B of the VQ-VAE-GAN step (bank on / off, twice)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vqvae.py -m gpu -x -q -k "step or bank or wn_coupling or generator or discriminator or checkpoint or two_ranks" 2>&1 | tail -8 > gpurun_out/bank_tests.txt
cat gpurun_out/bank_tests.txt
for rep in 1 2; do
  for v in 1 0; do
    TTTS_WN_BANK=$v timeout 200 python tools/vqvae_bench.py 32 8 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TTTS_WN_BANK=$v', round(d['ms_per_step'], 2), d['losses'])" | tee -a gpurun_out/bank_ab.txt
  done
done
