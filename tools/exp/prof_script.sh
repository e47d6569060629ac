#!/bin/bash
# rocprofv3 kernel stats of a small script: tools/exp/prof_script.sh tools/exp/<script>.py  -> top kernels
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf /tmp/ps_prof
(cd /tmp && timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_prof -o p -- python $R/$1 > /tmp/ps_out.txt 2>&1)
tail -6 /tmp/ps_out.txt | cut -c1-160
f=$(find /tmp/ps_prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s %6d calls  avg %9.1f us  total %8.2f ms" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
