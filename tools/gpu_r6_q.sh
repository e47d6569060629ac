#!/bin/bash
# kernel trace of the replayed VQ-VAE-GAN step -> overlap analysis (what is on the critical path?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r6q; mkdir -p $O
rm -rf /tmp/kt; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o g -- python $R/tools/exp/capture_debug.py 32 > $O/graph_run.txt 2>&1)
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); grep CAPTURE-OK $O/graph_run.txt
ms=$(grep CAPTURE-OK $O/graph_run.txt | sed 's/.*ms=\([0-9.]*\).*/\1/')
python tools/trace_overlap.py $f $(python -c "print(2*$ms)") 2 > $O/overlap_graph.txt; head -8 $O/overlap_graph.txt
