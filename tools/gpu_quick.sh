#!/bin/bash
# Quick GPU check used while iterating on the GPT path: kernel + GPT tests, then rocprofv3 kernel stats of the GPT bench.
#   bash tools/gpu_quick.sh <tag> ["pytest -k expression"]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/q_$1
mkdir -p $O
K=${2:-"attention or layernorm or colsum or batched or embed or gemm"}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "$K" -p no:cacheprovider > $O/kernels.log 2>&1; echo "KERNEL TESTS rc=$?"; tail -3 $O/kernels.log
timeout 600 python -m pytest tests/test_gpu_gpt.py -q -p no:cacheprovider > $O/gpt.log 2>&1; echo "GPT TESTS rc=$?"; tail -3 $O/gpt.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o gpt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vqvae > $O/bench.json 2> $O/prof.err); echo "PROF rc=$?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
python - "$O" <<'PY'
import csv, json, sys
o = sys.argv[1]
d = json.loads(open(o + "/bench.json").read().strip().splitlines()[-1])
print("GPT ms/step", d["ms_per_step"])
rows = list(csv.DictReader(open(o + "/kernel_stats.csv")))
for r in rows[:40]:
    if "spin_kernel" in r["Name"]: continue
    print("%-80s calls %5s avg %8.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
