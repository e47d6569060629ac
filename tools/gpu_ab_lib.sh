#!/bin/bash
# Same-box A/B of two builds of the library: tools/gpu_ab_lib.sh ALT.so  -> bench with the in-tree library and with ALT.so, back to back, twice.
# Build ALT.so first (here, before gpurun): e.g. tools/build_alt_lib.sh <git-rev> gemm.hip  -> ttts_amd/libttts_hip_alt.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ALT=${1:-ttts_amd/libttts_hip_alt.so}
for rep in 1 2; do
  for v in "" "$ALT"; do
    env TTTS_LIB="$v" timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-vqvae 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${v:-in-tree}', d['ms_per_step'], d['value'])"
  done
done
