#!/bin/bash
# Round-3 second closing run (after the NT GEMM dispatch change): full GPU suite, the default bench line, smoke, rocprofv3 kernel
# stats, FETCH / WRITE passes for profiles/pmc_traffic.json -- most important first, the GPU budget may cut the tail.  Output: gpurun_out/r3g/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r3g
mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/full.log 2>&1; echo "FULL rc=$?"; tail -4 $O/full.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "BENCH rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3g/bench.json").read().strip().splitlines()[-1])
print("GPT ms/step", d["ms_per_step"], "tok/s", d["value"], "roof", d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"])
print("kernels ms", d["roofline"]["all_kernels_ms_per_step"])
print("kernels TF", d["roofline"]["all_kernels_tflops"])
v = d.get("vqvae") or {}
print("vqvae", v.get("ms_per_step"), v.get("value"))
PY
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "SMOKE rc=$?"; tail -1 $O/smoke.log
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o gpt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vqvae > $O/prof_bench.json 2> $O/prof.err); echo "PROF rc=$?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/prof -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
bash tools/gpt_pmc.sh r03_pmc_gpt_fetch FETCH_SIZE > /dev/null
bash tools/gpt_pmc.sh r03_pmc_gpt_write WRITE_SIZE > /dev/null
echo DONE
