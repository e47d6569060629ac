#!/bin/bash
# thin / grouped kernels with batched requests: parity, the step, and their per-kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x 2>&1 | tail -2
for i in 1 2; do timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep CAPTURE-OK; done
bash tools/gpu_r6_u.sh > /dev/null 2>&1
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6u/kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if any(k in n for k in ['grouped','cout1','cin1','fused_taps_kernel<7','snake']):
        print(f"{n[:75]:75s} n/step={int(r['Calls'])/4:6.1f} avg={float(r['AverageNs'])/1e3:7.1f}us")
P
