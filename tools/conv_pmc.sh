#!/bin/bash
# PMC counters per kernel for one tools/conv_bench.py shape (kernel-trace + pmc only): tools/conv_pmc.sh "<shape substring>" COUNTER...
export TMPDIR=/tmp CB_B=${CB_B:-32} CB_ONLY="$1"
shift
rm -rf /tmp/cpmc
cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/cpmc -o p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py ${CB_FLAGS:-0} > /tmp/cpmc_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/cpmc/**/*counter_collection*.csv", recursive=True)
if not f:
    print("no counter csv"); print(open("/tmp/cpmc_stdout.txt").read()[-1500:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "ttts" not in k or "split" in k or "reduce" in k: continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print("    %-28s per launch %14.0f  (%d launches)" % (c, v / max(1, cnt[(k, c)]), cnt[(k, c)]))
PY
