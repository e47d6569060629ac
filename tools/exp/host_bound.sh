#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for B in 1 4; do
timeout 200 python tools/vqvae_bench.py $B 8 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B', {k: round(d[k], 2) for k in ('ms_per_step','host_issue_ms_in_loop')})"
done
