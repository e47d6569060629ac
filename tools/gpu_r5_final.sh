#!/bin/bash
# Round-5 closing run: full GPU test suite, smoke, rocprofv3 kernel stats of the GPT bench, the full default bench line.
# Output: gpurun_out/r5f/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r5f
mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/full.log 2>&1; echo "FULL rc=$?"; tail -4 $O/full.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "SMOKE rc=$?"; tail -1 $O/smoke.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o gpt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vqvae --no-diffusion > $O/prof_bench.json 2> $O/prof.err); echo "PROF rc=$?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/prof -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "BENCH rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5f/bench.json").read().strip().splitlines()[-1])
print("GPT ms/step", d["ms_per_step"], "median", d.get("ms_per_step_median"), "tok/s", d["value"], "roof", d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], "step_frac", d["roofline"].get("step_frac"))
print("cpu", {k: (v if not isinstance(v, dict) else v.get("value")) for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
v = d.get("vqvae") or {}
print("vqvae", v.get("ms_per_step"), v.get("ms_per_step_eager_streams"), v.get("ms_per_step_graph_replay"), v.get("value"), (v.get("roofline") or {}).get("frac"))
f = d.get("diffusion") or {}
print("diffusion", f.get("ms_per_step"), f.get("value"))
for r in d.get("hbm_kernels", []):
    print("  ", r)
PY
