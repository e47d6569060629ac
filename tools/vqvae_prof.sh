#!/bin/bash
# per-kernel times of the VQ-VAE-GAN step: tools/vqvae_prof.sh [B] [steps] -> gpurun_out/vqvae_kernel_stats.csv + top list
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vprof -o v -- python $GRAFT_REPO_ROOT/tools/vqvae_bench.py ${1:-32} ${2:-3} 1 > /tmp/vprof_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -1 /tmp/vprof_stdout.txt | cut -c1-150
f=$(find /tmp/vprof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/vqvae_kernel_stats.csv
python - "$f" ${2:-3} <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) + 1
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step %.1f ms, launches per step %d" % (tot / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:45]:
    print("%-84s %6d calls/step  %7.2f ms/step  avg %8.1f us" % (r["Name"][:84], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6, float(r["AverageNs"]) / 1e3))
PY
