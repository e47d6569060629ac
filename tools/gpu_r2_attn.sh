#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
timeout 150 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gpt.py -q -k "attn or attention or train_steps or tiny or full_config or dropout or grouped" > $O/t.log 2>&1; echo "TESTS rc=$?"; tail -4 $O/t.log
timeout 100 python bench.py --no-vqvae --no-cpu-baseline --steps 150 --warmup 10 > $O/b.json 2> $O/b.err; echo "B rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2a/b.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
PY
