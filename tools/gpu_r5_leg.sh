#!/bin/bash
# the full default bench line with the eager legs ahead of the GPT leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5q
timeout 600 python bench.py 2>gpurun_out/r5q/bench_reordered.err > gpurun_out/r5q/bench_reordered.json; echo rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5q/bench_reordered.json").read().strip().splitlines()[-1])
v = d["vqvae"]; f = d["diffusion"]
print("gpt", d["ms_per_step"], d["ms_per_step_median"], "vqvae", v["ms_per_step_eager_streams"], v["ms_per_step_graph_replay"], "frac", v["roofline"]["frac"], "diffusion", f["ms_per_step"], f["fp8_gemms"]["ms_per_step"])
print("cpu", d["cpu_baseline"]["value"], v["cpu_baseline"]["value"], f["cpu_baseline"]["value"])
PY
