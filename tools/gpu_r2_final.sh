#!/bin/bash
# Round-2 closing run: full GPU test suite, smoke, rocprofv3 kernel stats of the GPT bench, the full default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r2f
mkdir -p $O
timeout 200 python -m pytest tests -q -m gpu > $O/full.log 2>&1; echo "FULL rc=$?"; tail -6 $O/full.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "SMOKE rc=$?"; tail -1 $O/smoke.log
(cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o gpt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vqvae > $O/prof_bench.json 2> $O/prof.err); echo "PROF rc=$?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/prof -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
timeout 200 python bench.py > $O/bench.json 2> $O/bench.err; echo "BENCH rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2f/bench.json").read().strip().splitlines()[-1])
print("GPT ms/step", d["ms_per_step"], "tok/s", d["value"], "roof", d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"])
print("kernels ms", d["roofline"]["all_kernels_ms_per_step"])
v = d.get("vqvae") or {}
print("vqvae", v.get("ms_per_step"), v.get("value"))
PY
grep -E "ln_bwd_finalize|gpt_prepare_tokens|FillFunctor<long>|attn_delta" $O/kernel_stats.csv | cut -c1-160
