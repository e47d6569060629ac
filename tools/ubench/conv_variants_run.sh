#!/bin/bash
# runs every convolution library in tools/ubench/bin over tools/ubench/conv_cases.txt (baseline first); CONV_FLAGS = ttts_conv_ctx.flags
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B=tools/ubench/bin
mkdir -p gpurun_out
L=$(ls $B/conv_*.so | grep -v conv_base.so)
timeout 170 $B/conv_variants ${CONV_CASES:-tools/ubench/conv_cases.txt} $B/conv_base.so $L 2>&1 | tee gpurun_out/conv_variants.txt
