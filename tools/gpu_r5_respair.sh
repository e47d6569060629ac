#!/bin/bash
# fused ResBlock1 pairs (one autograd node per pair): parity tests, then the step with / without (same box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -x 2>&1 | tail -3
run() { env "$@" timeout 300 python tools/vqvae_bench.py 32 6 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'], 2), 'host', round(d['host_issue_ms_in_loop'], 1))"; }
for rep in 1 2; do run TTTS_RESPAIR=1; run TTTS_RESPAIR=0; done
