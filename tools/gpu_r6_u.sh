#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r6u; mkdir -p $O
rm -rf /tmp/vprof
(cd /tmp && TTTS_BRANCH_STREAMS=0 TTTS_D_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vprof -o v -- python $R/tools/vqvae_bench.py 32 3 1 > $O/run.txt 2>&1)
f=$(find /tmp/vprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
tail -1 $O/run.txt | cut -c1-200
