#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r6ai; mkdir -p $O
rm -rf /tmp/dprof; (cd /tmp && DFB_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof -o d -- python $R/tools/diffusion_bench.py > $O/diff.txt 2>&1)
f=$(find /tmp/dprof -name "*kernel_stats.csv" | head -1); cp "$f" $O/diffusion_kernel_stats.csv; tail -1 $O/diff.txt | cut -c1-100
