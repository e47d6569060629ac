#!/bin/bash
# round 6, fourth call: (1) world-1 RCCL vqvae log, (2) diffusion kernel stats with the fused attention, (3) GPT attention grouping /
# occupancy A/B inside the step (bench.py --no-vqvae --no-diffusion; attention families from roofline.all_kernels_ms_per_step)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 env TTTS_DP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 python tools/dp_world1_nccl.py vqvae > $O/w1_vqvae.log 2>&1; echo "vqvae rc $?"; grep -v "NCCL WARN\|^$" $O/w1_vqvae.log | tail -12
timeout 600 python -m pytest tests/test_gpu_diffusion.py -q -p no:cacheprovider -x 2>&1 | tail -3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_diff -o diff -- python $GRAFT_REPO_ROOT/tools/diffusion_bench.py > $GRAFT_REPO_ROOT/$O/diff_prof.log 2>&1)
f=$(find $O/prof_diff -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/diffusion_kernel_stats.csv && head -16 $O/diffusion_kernel_stats.csv | cut -c1-150
rm -rf $O/prof_diff
for m in 1 2; do
  TTTS_ATTN_GROUPING=$m timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gpt.py -q -p no:cacheprovider -x -k "attention or attn or train_step or fixture" 2>&1 | tail -2
done
run() {  # grouping, pad
  TTTS_ATTN_GROUPING=$1 TTTS_ATTN_LDS_PAD=$2 timeout 300 python bench.py --no-vqvae --no-diffusion --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']
print('grouping $1 pad $2: step', d['ms_per_step'], 'median', d['ms_per_step_median'], {n:v for n,v in k.items() if 'attn' in n or 'attention' in n})"
}
run 0 0
run 1 0
run 2 0
run 0 16,0,0
run 0 32,32,32
run 1 32,32,32
run 2 32,32,32
run 1,0,0 0
run 0,1,0 0
run 0,0,1 0
run 0 0
