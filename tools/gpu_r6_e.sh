#!/bin/bash
# round 6, fifth call: fused attention test detail, graphed diffusion step (test + eager/graph timing + kernel stats), world-1 vqvae
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_diffusion.py -q -p no:cacheprovider -x -k "fused or graphed" 2>&1 | grep -v "^$" | tail -45
timeout 600 env TTTS_DP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 python tools/dp_world1_nccl.py vqvae > $O/w1_vqvae.log 2>&1; echo "vqvae rc $?"; grep "vqvae-ok\|Error\|assert" $O/w1_vqvae.log | tail -5
for gmode in 0 1; do
  echo "DFB_GRAPH=$gmode"; DFB_GRAPH=$gmode DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1
done
echo "DFB_GRAPH=1 unfused"; TTTS_DIFFUSION_FUSED_ATTN=0 DFB_GRAPH=1 DFB_STEPS=20 timeout 300 python tools/diffusion_bench.py 2>&1 | tail -1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof -o d -- python $GRAFT_REPO_ROOT/tools/diffusion_bench.py > /tmp/dprof_stdout.txt 2>&1)
f=$(find /tmp/dprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/diffusion_kernel_stats.csv && head -14 $O/diffusion_kernel_stats.csv | cut -c1-150
