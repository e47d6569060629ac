#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6aa; mkdir -p $O
timeout 300 python tools/exp/aten_census.py 32 vqvae > $O/aten_vqvae.txt 2>&1; grep -A45 "^ATEN" $O/aten_vqvae.txt | cut -c1-200
TTTS_CONV_PRECISION=tf32class timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
TTTS_CONV_PRECISION=tf32class timeout 300 python tools/vqvae_bench.py 32 10 3 2>/dev/null | tail -1 | cut -c1-160
