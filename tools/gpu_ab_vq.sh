#!/bin/bash
# same-box A/B of a VQ-VAE-GAN step switch: tools/gpu_ab_vq.sh ENV_NAME [pytest -k expression] [A B]
#   -> step ms with ENV_NAME=A / B (default 1 / 0), twice, after the selected GPU tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=${3:-1}; Bv=${4:-0}
if [ -n "$2" ]; then
  timeout 900 python -m pytest tests/test_gpu_vqvae.py -m gpu -x -q -k "$2" 2>&1 | tail -15 | tee gpurun_out/ab_vq_tests.txt
fi
for rep in 1 2; do
  for v in $A $Bv; do
    vv=$v; [ "$v" == "default" ] && vv=""
    env $1=$vv timeout 200 python tools/vqvae_bench.py 32 8 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', round(d['ms_per_step'], 2), d['losses']['loss_gen_all'], d['max_mem_gb'], d['launch_batching']['slabs'])" | tee -a gpurun_out/ab_vq_$1.txt
  done
done
