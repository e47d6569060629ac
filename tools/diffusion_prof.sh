#!/bin/bash
# per-kernel times of the diffusion train step (tools/diffusion_bench.py) -> gpurun_out/diffusion_kernel_stats.csv + top list
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof -o d -- python $GRAFT_REPO_ROOT/tools/diffusion_bench.py > /tmp/dprof_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -1 /tmp/dprof_stdout.txt | cut -c1-200
f=$(find /tmp/dprof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/diffusion_kernel_stats.csv
python - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over the run" % (tot / 1e6))
for r in rows[:28]:
    print("%-84s %6s calls  %5.1f %%  avg %8.1f us" % (r["Name"][:84], r["Calls"], 100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3))
PY
