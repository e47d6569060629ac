#!/bin/bash
# per-kernel times of tools/conv_bench.py for one shape: tools/conv_prof.sh "<shape substring>" [flags]
mkdir -p gpurun_out
export TMPDIR=/tmp CB_B=${CB_B:-32} CB_ONLY="$1"
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof -o c -- python $GRAFT_REPO_ROOT/tools/conv_bench.py ${2:-0} > /tmp/cprof_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -2 /tmp/cprof_stdout.txt | cut -c1-170
f=$(find /tmp/cprof -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-86s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
