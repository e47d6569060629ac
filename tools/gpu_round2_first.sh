#!/bin/bash
# First GPU call of the next round (~3 GPU-min): everything that was written at the end of round 1 without GPU time to measure.
#   1. opt-in parity tests of kernels behind debug flags (TTTS_EXPERIMENTAL=1)
#   2. attention kernel bench: default vs the 64-query x 128-key splits (forward: flag 65536 / 131072, dQ: flag 262144)
#   3. bench.py with the dQ split on, to see the step-level effect
# Output: gpurun_out/round2_first.log
mkdir -p gpurun_out
{
echo "=== experimental parity tests"
TTTS_EXPERIMENTAL=1 timeout 120 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "kv2" 2>&1 | tail -4
echo "=== attention kernel bench (default | fwd kv2 | bwd with dq-kv2)"
KB_REPS=20 KB_ROUNDS=3 timeout 120 python tools/kernel_bench.py attn 2>&1 | tail -12
echo "=== bench, default"
timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
echo "=== bench, dQ work split (flag 262144)"
TTTS_DEBUG_FLAGS=262144 timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
echo "=== bench, dK/dV work split (flag 524288)"
TTTS_DEBUG_FLAGS=524288 timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
echo "=== bench, forward forced to the 128-query kernel (flag 131072)"
TTTS_DEBUG_FLAGS=131072 timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
} > gpurun_out/round2_first.log 2>&1
cat gpurun_out/round2_first.log
