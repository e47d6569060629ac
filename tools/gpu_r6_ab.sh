#!/bin/bash
# full GPU suite + the default bench line on the current code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6ab; mkdir -p $O
timeout 3000 python -m pytest tests -q -p no:cacheprovider -x -m gpu 2>&1 | tail -4
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 700 $O/bench.json
