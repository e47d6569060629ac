#!/bin/bash
# rocprofv3 kernel stats of the GPT bench for the in-tree library and an alternate one on the SAME box -> gpurun_out/prof/kernel_stats_{new,alt}.csv
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
for v in new alt; do
  L=""; [ $v == alt ] && L=$R/${1:-ttts_amd/libttts_hip_alt.so}
  rm -rf /tmp/prof_$v
  (cd /tmp && TTTS_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o trace -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-vqvae > $R/gpurun_out/bench_prof_$v.json 2> $R/gpurun_out/bench_prof_$v.err)
  for f in $(find /tmp/prof_$v -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/prof/kernel_stats_$v.csv; done
done
