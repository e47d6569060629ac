#!/bin/bash
# round-6 closing run: the driver's own sequence (GPU tests in collection order, smoke, default bench line) on the final code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6final; mkdir -p $O
timeout 3000 python -m pytest tests -q -p no:cacheprovider -x -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
