#!/bin/bash
# kernel trace of the replayed / eager VQ-VAE-GAN step -> overlap analysis (what is on the critical path?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r6q; mkdir -p $O
rm -rf /tmp/kt; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o g -- python $R/tools/exp/capture_debug.py 32 > $O/graph_run.txt 2>&1)
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); grep CAPTURE-OK $O/graph_run.txt
python tools/trace_overlap.py $f 230 2 > $O/overlap_graph.txt; head -12 $O/overlap_graph.txt
rm -rf /tmp/kt2; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o e -- python $R/tools/vqvae_bench.py 32 6 2 > $O/eager_run.txt 2>&1)
f=$(find /tmp/kt2 -name "*kernel_trace.csv" | head -1); tail -1 $O/eager_run.txt | cut -c1-150
python tools/trace_overlap.py $f 235 2 > $O/overlap_eager.txt; head -8 $O/overlap_eager.txt
