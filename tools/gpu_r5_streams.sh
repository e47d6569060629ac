#!/bin/bash
# Stream-count sweep of the VQ-VAE-GAN step now that the step is no longer host-bound (round 5: the mid-step host read-back is gone).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5ab; mkdir -p $O
run() { env "$@" timeout 300 python tools/vqvae_bench.py 32 6 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'], 2), 'host', round(d['host_issue_ms_in_loop'], 1))"; }
for rep in 1 2; do
  run A=default
  run TTTS_D_STREAMS=6
  run TTTS_BRANCH_STREAMS=4
  run TTTS_D_STREAMS=6 TTTS_BRANCH_STREAMS=4
  run TTTS_WGRAD_STREAMS=2
  run TTTS_D_STREAMS=6 TTTS_WGRAD_STREAMS=2
  run TTTS_D_STREAMS=0 TTTS_BRANCH_STREAMS=0
done 2>&1 | tee $O/streams_sweep.txt
