#!/bin/bash
# one-pass 1 x 1 weight gradient for the diffusion model's 64 .. 192-tile layers, re-measured with the XCD-contiguous tile order
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do for lib in "" ttts_amd/libttts_hip_w1_64.so ttts_amd/libttts_hip_w1all.so; do
  echo -n "lib=${lib:-in-tree}  "; TTTS_LIB=$lib DFB_STEPS=30 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-60
done; done
