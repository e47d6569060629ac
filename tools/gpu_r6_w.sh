#!/bin/bash
# GPT: GEMM epilogues with prefetched operands -- parity + same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_kernels.py -q -p no:cacheprovider -x 2>&1 | tail -3
bash tools/gpu_ab_lib.sh
