#!/bin/bash
# round 6, seventh call: where does the graphed diffusion step crash (faulthandler), per-shape 1 x 1 timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6g; mkdir -p $O
timeout 300 python -X faulthandler -m pytest tests/test_gpu_diffusion.py -q -p no:cacheprovider -x -k "graphed" > $O/graphed_test.log 2>&1; echo "rc $?"
grep -n "Fatal\|File \"/root/repo\|File \"/tmp/code\|Error\|passed\|failed" $O/graphed_test.log | head -40
timeout 300 python tools/conv1x1_bench.py 2>&1 | grep -v "^$\|amdgpu.ids" | tail -14
