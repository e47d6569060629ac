#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python tool: tools/gpu_prof_cmd.sh <name> <script> [args]; CSV -> gpurun_out/prof/<name>_kernel_stats.csv
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
NAME=$1; shift
SCRIPT=$1; shift
cd /tmp && rm -rf /tmp/prof_cmd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cmd -o trace -- python $GRAFT_REPO_ROOT/$SCRIPT "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof/${NAME}_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find /tmp/prof_cmd -name "*kernel_stats*.csv"); do cp $f gpurun_out/prof/${NAME}_kernel_stats.csv; done
head -12 gpurun_out/prof/${NAME}_kernel_stats.csv | cut -c1-160
tail -2 gpurun_out/prof/${NAME}_stdout.txt
