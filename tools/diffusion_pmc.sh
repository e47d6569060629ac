#!/bin/bash
# HBM-side traffic of the diffusion train step (all kernels): two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# tools/diffusion_bench.py; prints bytes per step and writes gpurun_out/diffusion_pmc_traffic.json
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=${1:-3}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/dpmc_$C
  (cd /tmp && DFB_STEPS=$STEPS DFB_WARMUP=1 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/dpmc_$C -o p -- python $GRAFT_REPO_ROOT/tools/diffusion_bench.py > /tmp/dpmc_$C.txt 2>&1)
done
python - $STEPS <<'PY'
import csv, glob, json, sys, collections
steps = int(sys.argv[1]) + 1          # + the warm-up step
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/dpmc_%s/**/*counter_collection*.csv" % c, recursive=True)
    if not f:
        print("no counter csv for", c); print(open("/tmp/dpmc_%s.txt" % c).read()[-800:]); continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0][:70]] += float(r["Counter_Value"])
    out[c] = {"all_kernels_kb_per_step": sum(agg.values()) / steps, "top": sorted(((v / steps, k) for k, v in agg.items()), reverse=True)[:12]}
fetch = out.get("FETCH_SIZE", {}).get("all_kernels_kb_per_step", 0.0) * 2 * 1024      # gfx950: 128-B read requests tallied at 64 B
write = out.get("WRITE_SIZE", {}).get("all_kernels_kb_per_step", 0.0) * 1024
out["bytes_per_step"] = fetch + write
out["_note"] = "units KB as reported by rocprofv3; FETCH_SIZE doubled per MI355X_MICROARCH.md; every kernel of the step"
json.dump(out, open("gpurun_out/diffusion_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != "top"}) for k, v in out.items()}))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for v, k in out.get(c, {}).get("top", [])[:6]:
        print("  %-10s %10.0f KB/step  %s" % (c, v, k))
PY
