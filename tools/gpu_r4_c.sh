#!/bin/bash
# round 4: vqvae suite after the stream work; GPT parity with the dW split overlap on; same-box A/B of the GPT step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vqvae.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4c_vqvae_tests.txt
TTTS_DW_SPLIT_OVERLAP=1 timeout 900 python -m pytest tests/test_gpu_gpt.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4c_gpt_tests.txt
bash tools/gpu_ab.sh TTTS_DW_SPLIT_OVERLAP 2>&1 | tee gpurun_out/r4c_ab_dw_overlap.txt
