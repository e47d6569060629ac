#!/bin/bash
# round 6, sixth call: conv1x1 kernel tests + whole vqvae / diffusion suites, reworked relpos attention, graphed diffusion step with logs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "conv1x1" 2>&1 | grep -v "^$" | tail -12
timeout 900 python -m pytest tests/test_gpu_diffusion.py -q -p no:cacheprovider -x 2>&1 | grep -v "^$" | tail -30
timeout 1800 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_fullsize.py tests/test_gpu_fp8.py tests/test_gpu_peq.py -q -p no:cacheprovider -x 2>&1 | grep -v "^$" | tail -8
echo "DFB eager"; DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1
echo "DFB eager, no conv1x1"; TTTS_DEBUG_FLAGS=512 DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1
echo "DFB graph"; DFB_GRAPH=1 DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py > $O/dfb_graph.log 2>&1; tail -1 $O/dfb_graph.log | cut -c1-300; grep -n "Error\|error\|failed" $O/dfb_graph.log | head -10
timeout 300 python tools/vqvae_bench.py 32 8 2 2>/dev/null | tail -1 | cut -c1-300
TTTS_DEBUG_FLAGS=512 timeout 300 python tools/vqvae_bench.py 32 8 2 2>/dev/null | tail -1 | cut -c1-300
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof -o d -- python $GRAFT_REPO_ROOT/tools/diffusion_bench.py > /tmp/dprof_stdout.txt 2>&1)
f=$(find /tmp/dprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/diffusion_kernel_stats.csv && head -14 $O/diffusion_kernel_stats.csv | cut -c1-150
