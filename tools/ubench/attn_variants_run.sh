#!/bin/bash
# runs every attention library in tools/ubench/bin (baseline first)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B=tools/ubench/bin
mkdir -p gpurun_out
L=$(ls $B/attn_*.so | grep -v attn_base.so)
timeout 170 $B/attn_variants $B/attn_base.so $L 2>&1 | tee gpurun_out/attn_variants.txt
