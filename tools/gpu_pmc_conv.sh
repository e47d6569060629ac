#!/bin/bash
# PMC counters for the conv micro-benchmark (separate pass from timing; --kernel-trace only, as required)
export TMPDIR=/tmp CB_B=32
cd /tmp && rm -rf /tmp/pmc_conv && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_conv -o pmc -- python $GRAFT_REPO_ROOT/tools/conv_bench.py 0 > /tmp/pmc_conv_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_conv/*counter_collection*.csv")
if not f:
    print("no counter csv"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "ttts" not in k: continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print("    %-28s %16.0f  (per launch %14.0f)" % (c, v, v / max(1, cnt[(k, c)])))
PY
