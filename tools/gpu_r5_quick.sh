#!/bin/bash
# quick check after a kernel change: the VQ-VAE / full-size / diffusion GPU tests, the stand-alone step and the bench legs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_fullsize.py tests/test_gpu_diffusion.py tests/test_gpu_fp8.py -q -p no:cacheprovider -x 2>&1 | tail -3
timeout 300 python tools/vqvae_bench.py 32 8 2 2>/dev/null | tail -1 | cut -c1-200
timeout 400 python bench.py --no-cpu-baseline --steps 50 2>/dev/null > $O/bench.json; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5q/bench.json").read().strip().splitlines()[-1])
v = d["vqvae"]; f = d["diffusion"]
print("gpt", d["ms_per_step"], "vqvae", v["ms_per_step_eager_streams"], v["ms_per_step_graph_replay"], "frac", v["roofline"]["frac"], "one-stream", v["roofline"]["one_stream_step_ms"], "fam", v["roofline"]["ms_per_step"], "diffusion", f["ms_per_step"], f["fp8_gemms"]["ms_per_step"])
PY
