#!/bin/bash
# Runs the -m gpu suite in independent processes (a memory fault in one group must not hide the others).
mkdir -p gpurun_out/diag
run() { name=$1; shift; echo "=== $name" ; timeout 900 python -m pytest "$@" -m gpu -q --timeout 600 -rfE -p no:cacheprovider 2>&1 | tail -60; echo "=== $name exit ${PIPESTATUS[0]}"; }
{
run probe_gemm tests/test_gpu_kernels.py -k "probe or gemm or colsum"
run rowwise tests/test_gpu_kernels.py -k "layernorm or embed or cross_entropy or adamw"
run attention tests/test_gpu_kernels.py -k "attention"
run vq_stft tests/test_gpu_kernels.py -k "vq or stft"
run gpt tests/test_gpu_gpt.py
run vqvae tests/test_gpu_vqvae.py
run peq tests/test_gpu_peq.py
run decode tests/test_gpu_decode.py
run diffusion tests/test_gpu_diffusion.py
} > gpurun_out/pytest_gpu.log 2>&1
grep -E "^===|passed|failed|error" gpurun_out/pytest_gpu.log | tail -40
