#!/bin/bash
# One-call check of the grouped / overlapped weight-gradient GEMMs: full GPU test suite, smoke, GPT bench with and without grouping.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2g
mkdir -p $O
timeout 260 python -m pytest tests -q -m gpu > $O/full.log 2>&1; echo "FULL rc=$?"
tail -15 $O/full.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "SMOKE rc=$?"; tail -2 $O/smoke.log
TTTS_GROUPED_DW=1 timeout 120 python bench.py --no-vqvae --no-cpu-baseline --steps 150 --warmup 10 > $O/b1.json 2> $O/b1.err; echo "B1 rc=$?"
TTTS_GROUPED_DW=0 timeout 120 python bench.py --no-vqvae --no-cpu-baseline --steps 150 --warmup 10 > $O/b0.json 2> $O/b0.err; echo "B0 rc=$?"
python - <<'PY'
import json
for n in ("b1", "b0"):
    try:
        d = json.loads(open("gpurun_out/r2g/%s.json" % n).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, "ms/step", d["ms_per_step"], "tok/s", d["value"], "dominant", r["kernel"], r["achieved"], "TF/s")
        print("   ms:", r["all_kernels_ms_per_step"])
        print("   TF:", r["all_kernels_tflops"])
    except Exception as e:
        print(n, "unreadable:", e)
PY
