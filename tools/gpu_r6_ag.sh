#!/bin/bash
# register-resident GroupNorm kernels: parity + diffusion-step A/B against HEAD's build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fp8.py -q -p no:cacheprovider -x 2>&1 | tail -2
for rep in 1 2; do for lib in "" ttts_amd/libttts_hip_alt.so; do
  for g in 0 1; do echo "diffusion lib=${lib:-in-tree} graph=$g"; TTTS_LIB=$lib DFB_GRAPH=$g DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-60; done
done; done
