#!/bin/bash
# Round 3, GPU call A: the head_dim-64 attention kernels (attn_dh64.hip) -- parity tests, A/B timing against the round-2
# kernels (TTTS_ATTN_OLD*), per-kernel rocprof table; then the round-2 leftovers that are still open (NT split GEMM,
# conv weight-gradient LATE, VQ unguarded).  Output: gpurun_out/r3a/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
echo "=== attention tests (new kernels)"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention or attn or dropout or reentrant" -p no:cacheprovider > $O/attn_tests.log 2>&1; echo "rc=$?"; tail -15 $O/attn_tests.log
echo "=== attention bench: new"
timeout 120 python tools/kernel_bench.py attn > $O/kb_attn_new.json 2>$O/kb_attn_new.err; cat $O/kb_attn_new.json
echo "=== attention bench: old"
TTTS_ATTN_OLD=1 TTTS_ATTN_OLD_DKDV=1 TTTS_ATTN_OLD_DQ=1 timeout 120 python tools/kernel_bench.py attn > $O/kb_attn_old.json 2>$O/kb_attn_old.err; cat $O/kb_attn_old.json
echo "=== per-kernel (rocprof, new kernels)"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_attn -o t -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py attn > /dev/null 2>&1)
for f in $(find /tmp/prof_attn -name "*kernel_stats*.csv"); do cp $f $O/attn_kernel_stats.csv; done
head -12 $O/attn_kernel_stats.csv | cut -c1-180
echo "=== GPT parity tests with the new kernels"
timeout 400 python -m pytest tests/test_gpu_gpt.py -q -x -p no:cacheprovider > $O/gpt_tests.log 2>&1; echo "rc=$?"; tail -5 $O/gpt_tests.log
echo "=== GPT bench (new attention)"
timeout 200 python bench.py --no-vqvae --no-cpu-baseline --steps 150 --warmup 10 > $O/bench_new.json 2> $O/bench_new.err; python -c "
import json; d=json.loads(open('$O/bench_new.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], d['roofline'].get('all_kernels_ms_per_step'))"
