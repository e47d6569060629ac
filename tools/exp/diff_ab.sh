#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_diffusion.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
  for v in 1 0; do
    TTTS_WSPLIT_CACHE=$v TTTS_WGRAD_ARENA=$v timeout 200 python tools/diffusion_bench.py 2>/dev/null | tail -1 | cut -c1-160 | sed "s/^/cache+arena=$v /"
  done
done
