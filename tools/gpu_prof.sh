#!/bin/bash
# rocprofv3 kernel trace + stats (CSV) of the headline bench mode; summary copied to gpurun_out/prof/
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
MODE=${1:-train}
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --mode $MODE > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$MODE.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$MODE.err
cd $GRAFT_REPO_ROOT
find /tmp/prof_out -type f | head -20
for f in $(find /tmp/prof_out -name "*kernel_stats*.csv"); do cp $f gpurun_out/prof/kernel_stats_$MODE.csv; done
head -40 gpurun_out/prof/kernel_stats_$MODE.csv | cut -c1-200
cat gpurun_out/bench_prof_$MODE.json | cut -c1-300
