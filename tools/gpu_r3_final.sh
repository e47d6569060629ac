#!/bin/bash
# Round-3 closing run: full GPU test suite, smoke, rocprofv3 kernel stats of the GPT bench, PMC passes (SQ for the attention
# kernels; FETCH / WRITE for profiles/pmc_traffic.json), the full default bench line.   Output: gpurun_out/r3f/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r3f
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/full.log 2>&1; echo "FULL rc=$?"; tail -4 $O/full.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "SMOKE rc=$?"; tail -1 $O/smoke.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o gpt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vqvae > $O/prof_bench.json 2> $O/prof.err); echo "PROF rc=$?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/prof -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
bash tools/gpt_pmc.sh r03_pmc_gpt_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE > /dev/null
bash tools/gpt_pmc.sh r03_pmc_gpt_sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES > /dev/null
bash tools/gpt_pmc.sh r03_pmc_gpt_fetch FETCH_SIZE > /dev/null
bash tools/gpt_pmc.sh r03_pmc_gpt_write WRITE_SIZE > /dev/null
grep -A9 "dh64" gpurun_out/pmc/r03_pmc_gpt_sq.txt | head -34
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "BENCH rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3f/bench.json").read().strip().splitlines()[-1])
print("GPT ms/step", d["ms_per_step"], "tok/s", d["value"], "roof", d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"])
print("kernels ms", d["roofline"]["all_kernels_ms_per_step"])
print("cpu", {k: (v if not isinstance(v, dict) else v.get("value")) for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
v = d.get("vqvae") or {}
print("vqvae", v.get("ms_per_step"), v.get("value"))
PY
