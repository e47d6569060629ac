#!/bin/bash
# round 6: tf32class / loss-scale tests after dropping the subnormal counter, then the default bench line (legs reordered)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vqvae.py -q -p no:cacheprovider -x -k "tf32class or dynamic_loss or nan" 2>&1 | grep -v "^$" | tail -3
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6j/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in d if k.startswith(("vqvae_", "diffusion_", "gpt_"))})
v = d["vqvae"]; print("vqvae", v["ms_per_step"], v["ms_per_step_eager_streams"], v["ms_per_step_graph_replay"], v["roofline"]["frac"], v["roofline"]["families_ms"], "tf32", v["tf32class"]["ms_per_step"], v["tf32class"]["loss_scale"], v["tf32class"]["f16_saturated"], v["tf32class"]["f16_flushed"], v["losses"], v["tf32class"]["losses"])
f = d["diffusion"]; print("diffusion", f["ms_per_step"], f["ms_per_step_eager"], f["ms_per_step_graph_replay"], f["graphs_recorded"], f["roofline"]["frac"], "fp8", f["fp8_gemms"]["ms_per_step_eager"], f["fp8_gemms"]["ms_per_step_graph_replay"], f["fp8_gemms"]["with_tf32class_convs"])
PY
