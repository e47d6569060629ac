#!/bin/bash
# PMC counters per kernel for the GPT forward+backward (kernel-trace + pmc only): tools/gpt_pmc.sh OUTNAME COUNTER...
# prints per-launch averages for the GEMM / attention kernels into gpurun_out/pmc/OUTNAME.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1
shift
mkdir -p $R/gpurun_out/pmc
rm -rf /tmp/gpmc
cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/gpmc -o p -- python $R/tools/gpt_step_once.py > /tmp/gpmc_stdout.txt 2>&1
cd $R
python - > gpurun_out/pmc/$OUT.txt <<PY
import csv, glob, collections
f = glob.glob("/tmp/gpmc/**/*counter_collection*.csv", recursive=True)
if not f:
    print("no counter csv"); print(open("/tmp/gpmc_stdout.txt").read()[-1500:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if not any(s in k for s in ("gemm", "attn")): continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print("    %-28s per launch %16.0f  (%d launches)" % (c, v / max(1, cnt[(k, c)]), cnt[(k, c)]))
PY
tail -5 /tmp/gpmc_stdout.txt >> gpurun_out/pmc/$OUT.txt
