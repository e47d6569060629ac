#!/bin/bash
# PMC counters per attention kernel (kernel-trace + pmc only) over tools/kernel_bench.py attn:
#   tools/attn_pmc.sh OUTNAME COUNTER...   -> gpurun_out/pmc/OUTNAME.txt (per-launch averages)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1
shift
mkdir -p $R/gpurun_out/pmc
rm -rf /tmp/apmc
cd /tmp && KB_REPS=3 KB_ROUNDS=1 timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/apmc -o p -- python $R/tools/kernel_bench.py attn > /tmp/apmc_stdout.txt 2>&1
cd $R
python - > gpurun_out/pmc/$OUT.txt <<PY
import csv, glob, collections
f = glob.glob("/tmp/apmc/**/*counter_collection*.csv", recursive=True)
if not f:
    print("no counter csv"); print(open("/tmp/apmc_stdout.txt").read()[-1500:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:72]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if "attn" not in k: continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print("    %-28s per launch %16.0f  (%d launches)" % (c, v / max(1, cnt[(k, c)]), cnt[(k, c)]))
PY
cat gpurun_out/pmc/$OUT.txt | grep -A12 "true>"
