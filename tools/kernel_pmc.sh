#!/bin/bash
# PMC counters of the kernels a small python snippet launches: tools/kernel_pmc.sh OUTNAME "python code" NAME_FILTER COUNTER...
# (kernel-trace + pmc only).  Per-launch averages -> gpurun_out/pmc/OUTNAME.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; CODE=$2; FILT=$3
shift 3
mkdir -p $R/gpurun_out/pmc
rm -rf /tmp/kpmc
printf 'import sys, torch\nsys.path.insert(0, "%s")\nfrom ttts_amd import ops\n%s\ntorch.cuda.synchronize()\n' "$R" "$CODE" > /tmp/kpmc_snippet.py
cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/kpmc -o p -- python /tmp/kpmc_snippet.py > /tmp/kpmc_stdout.txt 2>&1
cd $R
python - "$FILT" > gpurun_out/pmc/$OUT.txt <<PY
import csv, glob, collections, sys
f = glob.glob("/tmp/kpmc/**/*counter_collection*.csv", recursive=True)
if not f:
    print("no counter csv"); print(open("/tmp/kpmc_stdout.txt").read()[-1500:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:72]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if sys.argv[1] not in k: continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print("    %-28s per launch %16.0f  (%d launches)" % (c, v / max(1, cnt[(k, c)]), cnt[(k, c)]))
PY
cat gpurun_out/pmc/$OUT.txt
