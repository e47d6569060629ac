#!/bin/bash
# bench (both modes) + rocprofv3 kernel trace of the headline mode; outputs under gpurun_out/
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --mode graph_nodropout > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err )
ls -la gpurun_out/prof | head
find gpurun_out/prof -name "*stats*" | head
cat gpurun_out/bench_train.json gpurun_out/bench_graph.json gpurun_out/bench_prof.json
tail -3 gpurun_out/bench_train.err gpurun_out/bench_prof.err
