#!/bin/bash
# SQ counters of conv1x1_b3_kernel at the diffusion qkv shape (16 x 512 -> 1536 x 400)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r6aj; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT"; do
  i=$((i+1)); rm -rf /tmp/p1
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p1 -o p -- python $R/tools/exp/conv1x1_one.py 16 512 1536 400 6 > /dev/null 2>&1)
  python - <<PY > $O/sq$i.txt
import csv, glob, collections
f = glob.glob("/tmp/p1/**/*counter_collection*.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:50]
    if "conv1x1" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        print("    %-28s per launch %14.0f  (%d launches)" % (c, v / max(1, cnt[(k, c)]), cnt[(k, c)]))
PY
  cat $O/sq$i.txt
done
