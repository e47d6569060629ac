#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
hostname; rocm-smi --showclocks --showperflevel --showpowerprofile 2>&1 | grep -v "^=\|^$" | head -12
timeout 300 python tools/exp/capture_debug.py 32 2>&1 | grep "CAPTURE-OK\|Fatal\|Error\|failed" | head -3
timeout 300 python tools/vqvae_bench.py 32 6 2 2>/dev/null | tail -1 | cut -c1-200
timeout 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-vqvae --no-diffusion 2>/dev/null | tail -1 | cut -c1-200
(timeout 60 python tools/exp/capture_debug.py 32 > /dev/null 2>&1 &) ; sleep 35; rocm-smi --showclocks --showpower 2>&1 | grep -v "^=\|^$" | head -8
