#!/bin/bash
# Cycle-stamp build of the weights-in-registers NT kernel + its reader (tools/ubench/wreg_trace.cpp) into tools/ubench/bin/.
#   bash tools/ubench/wreg_trace_build.sh ; on the box: cd tools/ubench/bin && ./wreg_trace ./tr_full.so <epilogue 0|1> [N]
set -e
cd "$(dirname "$0")/../.."
B=tools/ubench/bin; mkdir -p $B
/opt/rocm/bin/hipcc -O2 -o $B/wreg_trace tools/ubench/wreg_trace.cpp -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Iinclude -Ittts_amd/csrc -DWR_TRACE=1 -shared -o $B/tr_full.so ttts_amd/csrc/gemm.hip ttts_amd/csrc/lib.hip
ls -la $B/wreg_trace $B/tr_full.so
