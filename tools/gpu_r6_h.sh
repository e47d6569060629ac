#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fp8.py -q -p no:cacheprovider -x 2>&1 | grep -v "^$" | tail -6
for gmode in 0 1; do echo "DFB_GRAPH=$gmode"; DFB_GRAPH=$gmode DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1 | cut -c1-250; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof -o d -- python $GRAFT_REPO_ROOT/tools/diffusion_bench.py > /tmp/dprof_stdout.txt 2>&1)
f=$(find /tmp/dprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/diffusion_kernel_stats.csv && grep "relattn" $O/diffusion_kernel_stats.csv | cut -c1-150
