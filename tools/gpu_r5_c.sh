#!/bin/bash
# Round-5 third run: fp8 GEMM (LDS-staged kernel, split weight gradient) tests + diffusion kernel stats in both precisions, kernel
# stats and HBM-traffic counters of the VQ-VAE-GAN step (final code), the diffusion bench leg.  Output: gpurun_out/r5c/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r5c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fp8.py -q -p no:cacheprovider -s > $O/fp8.log 2>&1; echo "FP8 rc=$?"; grep -E "fp8 diffusion|passed|failed|Error|error|assert " $O/fp8.log | tail -12
for mode in f32 fp8; do
  (cd /tmp && TTTS_DIFFUSION_PRECISION=$mode DFB_STEPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof_$mode -o d -- python $R/tools/diffusion_bench.py > $O/diff_$mode.txt 2>&1)
  tail -1 $O/diff_$mode.txt | cut -c1-300
  f=$(find /tmp/dprof_$mode -name "*kernel_stats.csv" | head -1); cp "$f" $O/diffusion_${mode}_kernel_stats.csv
done
python - <<'PY'
import csv
for mode in ("fp8",):
    rows = list(csv.DictReader(open("gpurun_out/r5c/diffusion_%s_kernel_stats.csv" % mode)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(mode, "total kernel time %.1f ms over 7 steps" % (tot / 1e6))
    for r in rows[:12]:
        print("  %-70s %5s calls %5.1f %% avg %8.1f us" % (r["Name"][:70], r["Calls"], 100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3))
PY
TTTS_BRANCH_STREAMS=0 TTTS_D_STREAMS=0 bash tools/vqvae_prof.sh 32 3 > $O/vqvae_prof.txt 2>&1; cp gpurun_out/vqvae_kernel_stats.csv $O/vqvae_kernel_stats.csv; head -30 $O/vqvae_prof.txt
TTTS_BRANCH_STREAMS=0 TTTS_D_STREAMS=0 bash tools/vqvae_pmc.sh 1 > $O/vqvae_pmc.txt 2>&1; cp gpurun_out/vqvae_pmc_traffic.json $O/vqvae_pmc_traffic.json; tail -20 $O/vqvae_pmc.txt
timeout 600 python bench.py --no-vqvae --no-cpu-baseline --steps 50 > $O/bench.json 2> $O/bench.err; echo "BENCH rc=$?"; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c/bench.json").read().strip().splitlines()[-1])
f = d.get("diffusion") or {}
print("diffusion", f.get("ms_per_step"), f.get("value"), "fp8:", (f.get("fp8_gemms") or {}).get("ms_per_step"), (f.get("fp8_gemms") or {}).get("loss"), "loss", f.get("loss"))
PY
