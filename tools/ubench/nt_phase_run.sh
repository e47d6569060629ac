#!/bin/bash
# runs every variant library in tools/ubench/bin (baseline first)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B=tools/ubench/bin
mkdir -p gpurun_out
L=$(ls $B/nt_*.so | grep -v nt_base.so)
timeout 170 $B/nt_phase $B/nt_base.so $L 2>&1 | tee gpurun_out/nt_phase.txt
