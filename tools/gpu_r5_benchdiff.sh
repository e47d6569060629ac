#!/bin/bash
# Why does bench.py's VQ-VAE-GAN leg read 132 ms when tools/vqvae_bench.py reads 125-127 ms on the same code?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5ab; mkdir -p $O
p() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); v = d.get('vqvae') or {}; print('$1', 'gpt', d['ms_per_step'], 'vqvae', v.get('ms_per_step_eager_streams'), v.get('ms_per_step_graph_replay'))"; }
{
TTTS_BENCH_GC_FREEZE=0 timeout 300 python bench.py --no-cpu-baseline --no-diffusion --steps 20 2>/dev/null | p "gc-freeze=0"
TTTS_BENCH_GC_FREEZE=1 timeout 300 python bench.py --no-cpu-baseline --no-diffusion --steps 20 2>/dev/null | p "gc-freeze=1"
timeout 300 python tools/vqvae_bench.py 32 8 2 2>/dev/null | tail -1 | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_fullsize.py tests/test_gpu_diffusion.py -q -p no:cacheprovider -x 2>&1 | tail -3
} 2>&1 | tee $O/benchdiff2.txt
