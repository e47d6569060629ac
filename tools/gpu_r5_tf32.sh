#!/bin/bash
# the single-pass "TF32-class" convolution mode: accuracy + step parity tests, then the bench line (which times the mode beside the default)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5q
timeout 600 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -s -k "tf32class" 2>&1 | grep -E "tf32class|passed|failed|Error|assert" | tail -8
timeout 600 python bench.py --no-cpu-baseline --no-diffusion --steps 50 2>gpurun_out/r5q/bench_tf32.err > gpurun_out/r5q/bench_tf32.json; echo rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5q/bench_tf32.json").read().strip().splitlines()[-1])
v = d["vqvae"]
print("gpt", d["ms_per_step"], "vqvae", v["ms_per_step_eager_streams"], v["ms_per_step_graph_replay"], "tf32class", v["tf32class"]["ms_per_step"], v["tf32class"]["losses"])
print("default losses", v["losses"])
PY
