#!/bin/bash
# Round 3, GPU call B: forward kernel variants (A/B), attention tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3b
mkdir -p $O
echo "=== attention tests"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention or attn or dropout or reentrant" -p no:cacheprovider > $O/attn_tests.log 2>&1; echo "rc=$?"; tail -4 $O/attn_tests.log
echo "=== attention bench (per-kernel via rocprof)"
(cd /tmp && rm -rf /tmp/prof_attn && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_attn -o t -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py attn > $GRAFT_REPO_ROOT/$O/kb.json 2>&1)
for f in $(find /tmp/prof_attn -name "*kernel_stats*.csv"); do cp $f $O/attn_kernel_stats.csv; done
grep attn $O/attn_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
cat $O/kb.json | tail -12
