#!/bin/bash
# Round-5 second run: the new tests (fp8 GEMMs, full-clip gradient parity, state / ordering fixes, self-spawning launchers),
# rocprofv3 kernel stats of the diffusion step in both precisions, the bench's GPT + diffusion legs.  Output: gpurun_out/r5b/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r5b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fp8.py -q -p no:cacheprovider -s > $O/fp8.log 2>&1; echo "FP8 rc=$?"; grep -E "fp8 diffusion|passed|failed|Error|error" $O/fp8.log | tail -12
timeout 900 python -m pytest "tests/test_gpu_fullsize.py::test_config3_gradients_at_full_clip_length_match_the_oracle" -q -p no:cacheprovider -s > $O/grad.log 2>&1; echo "GRAD rc=$?"; grep -E "config #3 gradients|passed|failed|Error|assert" $O/grad.log | tail -12
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_gpt.py -q -p no:cacheprovider -k "frozen or private or joins or spawns" > $O/new.log 2>&1; echo "NEW rc=$?"; tail -5 $O/new.log
for mode in f32 fp8; do
  (cd /tmp && TTTS_DIFFUSION_PRECISION=$mode DFB_STEPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof_$mode -o d -- python $R/tools/diffusion_bench.py > $O/diff_$mode.txt 2>&1)
  tail -1 $O/diff_$mode.txt | cut -c1-300
  f=$(find /tmp/dprof_$mode -name "*kernel_stats.csv" | head -1); cp "$f" $O/diffusion_${mode}_kernel_stats.csv
done
python - <<'PY'
import csv
for mode in ("f32", "fp8"):
    rows = list(csv.DictReader(open("gpurun_out/r5b/diffusion_%s_kernel_stats.csv" % mode)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(mode, "total kernel time %.1f ms over 7 steps" % (tot / 1e6))
    for r in rows[:14]:
        print("  %-70s %5s calls %5.1f %% avg %8.1f us" % (r["Name"][:70], r["Calls"], 100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3))
PY
timeout 600 python bench.py --no-vqvae --steps 100 > $O/bench.json 2> $O/bench.err; echo "BENCH rc=$?"; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5b/bench.json").read().strip().splitlines()[-1])
print("GPT ms/step", d["ms_per_step"], "median", d.get("ms_per_step_median"))
f = d.get("diffusion") or {}
print("diffusion", f.get("ms_per_step"), f.get("value"), "fp8:", (f.get("fp8_gemms") or {}).get("ms_per_step"), (f.get("fp8_gemms") or {}).get("loss"), "loss", f.get("loss"))
PY
