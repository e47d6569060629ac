#!/bin/bash
# A/B of an engine switch on one box: tools/gpu_ab.sh ENV_NAME  -> bench with ENV_NAME=1 and =0, back to back, twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
  for v in 1 0; do
    env $1=$v timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-vqvae 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', d['ms_per_step'])"
  done
done
