#!/bin/bash
# PMC counters for the micro-benchmarked kernels (separate pass from timing; --kernel-trace only, as required)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp KB_REPS=2 KB_ROUNDS=1 KB_QUICK=1
WHICH=${1:-gemm}
shift
cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_out -o pmc -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py $WHICH > /tmp/pmc_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/pmc_out -type f | head
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_out/*counter_collection*.csv")
if not f:
    print("no counter csv"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
print(rows[0].keys())
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "ttts" not in k: continue
    print(k)
    for c, v in agg[k].items():
        print("    %-28s %16.0f  (per launch %14.0f)" % (c, v, v / max(1, cnt[(k, c)])))
PY
