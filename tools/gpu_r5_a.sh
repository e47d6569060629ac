#!/bin/bash
# Round-5 first run: full GPU suite (with the new launch / ordering tests) + the default bench line.  Output: gpurun_out/r5a/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r5a
mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/full.log 2>&1; echo "FULL rc=$?"; tail -15 $O/full.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "SMOKE rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "BENCH rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5a/bench.json").read().strip().splitlines()[-1])
print("GPT ms/step", d["ms_per_step"], "median", d.get("ms_per_step_median"), "tok/s", d["value"], "roof", d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], "step_frac", d["roofline"].get("step_frac"))
print(d["roofline"]["all_kernels_ms_per_step"])
v = d.get("vqvae") or {}
print("vqvae", v.get("ms_per_step"), v.get("ms_per_step_eager_streams"), v.get("ms_per_step_graph_replay"), v.get("value"))
print("vqvae roof", {k: v_ for k, v_ in (v.get("roofline") or {}).items() if k not in ("traffic_note", "timing", "kernel")})
f = d.get("diffusion") or {}
print("diffusion", f.get("ms_per_step"), f.get("value"))
PY
